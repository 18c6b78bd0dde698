"""GPU-box helper for profiling: feature extraction alone (k_extract_agz, both row formats) over 16384 mid-game positions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import elf_amd
B = 16384
eng = elf_amd.GoEngine(19, B, 0)
seeds = torch.from_numpy((np.arange(B, dtype=np.uint64) * np.uint64(0x9E3779B9) + np.uint64(1)).view(np.int64)).cuda()
eng.playout(seeds, max_steps=120)
d4 = torch.arange(B, device="cuda", dtype=torch.int32) % 8
for fmt, nbytes in (("f32_nchw", 25992 + 736), ("f16_nhwc", 12996 + 736)):
    out = eng.extract_agz(None, d4, n=B, fmt=fmt)
    for _ in range(3):
        eng.extract_agz(None, d4, out=out, n=B, fmt=fmt)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20):
        eng.extract_agz(None, d4, out=out, n=B, fmt=fmt)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print("extract_agz %s x%d: %.1f us  %.0f GB/s algorithmic" % (fmt, B, dt * 1e6, B * nbytes / dt / 1e9), flush=True)
