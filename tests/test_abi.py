"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/elf_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "elf_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(elf(?:go|mcts|sp|net|train|rec|rq)_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    import elf_amd
    from elf_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "libelf_amd.so lacks %s" % n
        assert n in _lib.SIGNATURES, "elf_amd/_lib.py lacks a prototype for %s" % n
    assert set(_lib.SIGNATURES) == set(names)
    assert elf_amd.lib().elfgo_version().startswith(b"elf_amd")


def test_bad_arguments_are_status_codes_not_crashes(built):
    import elf_amd
    L = elf_amd.lib()
    h = ctypes.c_void_p()
    assert L.elfgo_create(13, 4, 0, None, ctypes.byref(h)) == -1  # null zobrist
    z = (ctypes.c_uint64 * 441)()
    assert L.elfgo_create(13, 4, 0, z, ctypes.byref(h)) == -2     # unsupported size
    assert L.elfgo_destroy(None) == -1
    assert b"bad argument" in L.elfgo_error_string(-1)


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "elf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "libgo_oracle" not in txt and "libelfref" not in txt, f


def test_mcts_bad_arguments_are_status_codes(built):
    import elf_amd
    L = elf_amd.lib()
    h = ctypes.c_void_p()
    assert L.elfmcts_create(None, 1, 64, 16, None, ctypes.byref(h)) == -1
    assert L.elfmcts_destroy(None) == -1
    assert L.elfsp_create(None, 0, None, ctypes.byref(h)) == -1
    assert L.elfsp_destroy(None) == -1
    assert L.elfsp_begin_step(None, None, 0, None, None) == -1
    # the trainer-side entry points added in round 4: a null store is a status code
    assert L.elftrain_set_keep_states(None, 1) == -1
    assert L.elftrain_put_async(None, 0, None, 0, ctypes.c_float(1.0), 0, None, 0, None, 0, None) == -1


def test_tree_memory_accounting_and_the_round_5_entry_points(built):
    """elfmcts_tree_bytes_per_game is host arithmetic: 5 888-B small records (19x19; 1 920 B at 9x9), nodes / 16 + 1 big records of
    11 520 B (3 200 B), the id arrays, the leaf / row tables and the path rows -- what a caller sizes num_games against; the
    per-thread draw accessors refuse null handles with a status code."""
    import elf_amd
    L = elf_amd.lib()
    for n, small, big in ((19, 5888, 11520), (9, 1920, 3200)):
        for nodes in (64, 8192, 33792):
            b = elf_amd.tree_bytes_per_game(n, nodes)
            records = nodes * small + (nodes // 16 + 1) * big
            assert records - big < b < records + (nodes + nodes // 16 + 1) * 13 + (1 << 17), (n, nodes, b)   # 13 B of id arrays per id
    assert elf_amd.tree_bytes_per_game(19, 8192) * 4096 < 230e9          # the search-only line: 4096 games fit 288 GB
    assert elf_amd.tree_bytes_per_game(13, 8192) == 0 and elf_amd.tree_bytes_per_game(19, 0) == 0
    # the second form takes what is laid out per leaf of a step and per search thread: 512 B of path row per leaf (1024 leaves = 512 KB)
    b1 = L.elfmcts_tree_bytes_per_game2(19, 8192, 1, 16, 1024)
    assert b1 == elf_amd.tree_bytes_per_game(19, 8192)
    b2 = L.elfmcts_tree_bytes_per_game2(19, 8192, 8, 128, 8 * 1024)
    assert b2 - b1 == (1024 - 16) * (512 + 9 * 4) + 7 * 1024 + 7 * 4
    assert L.elfmcts_tree_bytes_per_game2(19, 8192, 0, 16, 1024) == 0
    assert L.elfmcts_pool_info(None, None, 0) == -1 and L.elfmcts_count_live(None, None) == -1
    assert L.elfmcts_num_threads(None) == -1 and L.elfmcts_thread_draws(None, None, None) == -1
    assert L.elfsp_ts_requests_deferred(None) == -1


def test_selfplay_refuses_to_run_without_gpu(built):
    import pytest
    import torch
    import elf_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        elf_amd.SelfPlay(19, 2)


def test_engine_refuses_to_run_without_gpu(built):
    import pytest
    import torch
    import elf_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        elf_amd.GoEngine(19, 4)


def test_header_is_plain_c_and_links_from_c(built, tmp_path):
    """include/elf_amd.h compiles as C11 with -Wall -Werror -pedantic; a C program linked against libelf_amd.so runs the
    host-only entry points (record format helpers) without a GPU."""
    import subprocess
    exe = str(tmp_path / "abi_c_check")
    lib = os.path.join(ROOT, "elf_amd", "lib")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "native", "abi_c_check.c"), "-o", exe, "-L", lib, "-lelf_amd",
           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "abi_c_check ok" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])


def test_pmc_profiles_were_measured_on_the_kernel_sources_in_the_tree():
    """bench.py prices `roofline.traffic`, `frac_pmc` and the issue roofs against profiles/pmc_traffic.json / pmc_issue.json / valu_mix.json
    only while they carry the hash of the kernel sources in the tree (elf_amd._lib.kernel_source_hash); a kernel edit without a new
    tools/profile_all.sh visit would silently blank those fields in the driver's line."""
    import json
    from elf_amd._lib import kernel_source_hash
    h = kernel_source_hash()
    for name in ("pmc_traffic.json", "pmc_issue.json", "valu_mix.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d.get("_source", {}).get("kernel_source_hash") == h, name
