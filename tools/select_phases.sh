#!/bin/bash
# GPU-box tool: cycle attribution of k_mcts_select by phase. Builds a PROFILE variant of the library in place (the committed
# build is restored afterwards), runs the search-only bench and prints the accumulated s_memtime ticks per phase.
set -e
make -C elf_amd/csrc clean >/dev/null
make -C elf_amd/csrc HIPCC="/opt/rocm/bin/hipcc -DELF_PROFILE_SELECT" >/dev/null 2>&1
python - <<'PY'
import ctypes as C, json, sys
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--workload", "mcts", "--net", "random", "--features", "f16", "--games", "1024", "--groups", "1", "--nodes-per-game", "8192",
            "--rollouts", "2048", "--pregrow", "0", "--warmup", "24", "--steps", "32", "--no-cpu-baseline"]
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("search-only (profile build)", d["value"], d["config"]["select_ms"], "depth", d["config"]["mean_depth"])
import elf_amd
L = C.CDLL(elf_amd._lib.LIB_PATH)
out = (C.c_uint64 * 8)()
L.elfprof_select_phases(out)
names = ["header + scoring order arrive (round trip 1)", "gather, scores, reductions, FPU sum", "new node: id, header, order insertion", "parent slot -> LDS",
         "Board::forward", "slot store issue / loop exit", "leaf bookkeeping", "fence (stores visible)"]
tot = sum(out)
roll = 56 * 1024 * 16
print("  total %.0f ticks per rollout" % (tot / roll))
for n, v in zip(names, out):
    print("  %-46s %6.2f %%  %8.0f ticks/rollout" % (n, 100.0 * v / tot, v / roll))
st = (C.c_uint64 * 4096)()
L.elfprof_select_steps(st)
import numpy as np
a = np.array(st[:], dtype=np.float64).reshape(1024, 4)
a = a[a[:, 2] > 0]
mean = a[:, 0] / a[:, 2]
print("  per step (one launch): mean wave %.0f ticks, slowest wave %.0f ticks (x %.2f); visited nodes per wave: mean %.1f, most %.0f"
      % (mean[8:].mean(), a[8:, 1].mean(), (a[8:, 1] / mean[8:]).mean(), 16 * d["config"]["mean_depth"], a[8:, 3].mean()))
PY
make -C elf_amd/csrc clean >/dev/null
make -C elf_amd/csrc >/dev/null 2>&1
