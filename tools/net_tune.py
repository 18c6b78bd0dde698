"""GPU-box helper: exhaustive MIOpen tuning of the benchmark net's convolution shapes (BASELINE configs[2]: 20 blocks x 256
channels, fp16 channels_last, 2048 positions per call), writing MIOpen's user databases to the directory given as argv[1].
The tuned databases are committed under elf_amd/data/miopen_db/ and bench.py points MIOPEN_USER_DB_PATH at them, so a fresh
box runs the tuned kernels without searching.  Usage: python tools/net_tune.py <db_dir> [rows]"""
import os
import sys
import time

db = os.path.abspath(sys.argv[1])
os.makedirs(db, exist_ok=True)
os.environ["MIOPEN_USER_DB_PATH"] = db
os.environ.setdefault("MIOPEN_FIND_ENFORCE", "4")      # SEARCH_DB_UPDATE: tune every applicable solver, keep the results
os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from elf_amd.net import FusedInferenceNet, make_net  # noqa: E402

torch.backends.cudnn.benchmark = True
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
net = make_net(dtype=torch.float16, channels_last=True, fold_bn=True)
f = FusedInferenceNet(net)
s = (torch.rand(rows, 18, 19, 19, device="cuda") < 0.3).half().contiguous(memory_format=torch.channels_last)
t0 = time.time()
f({"s": s})
torch.cuda.synchronize()
print("first call (tuning): %.1f s" % (time.time() - t0), flush=True)
for _ in range(3):
    f({"s": s})
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    f({"s": s})
torch.cuda.synchronize()
dt = (time.time() - t0) / 10
fl = 2.0 * 361 * 9 * (18 * 256 + 40 * 256 * 256)
print("net call %d rows: %.2f ms  %.0f TFLOP/s" % (rows, dt * 1e3, fl * rows / dt / 1e12), flush=True)
print(sorted(os.listdir(db)))
