"""GPU-box helper: throughput of the 20x256 net under the three execution paths (eager / fused HIP epilogue / MIOpen fused ops),
and of the feature kernels in both row formats.  Usage: python tools/net_bench2.py [bs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import elf_amd
from elf_amd.net import make_net, FusedInferenceNet
torch.backends.cudnn.benchmark = True

def timeit(f, it=6):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.time() - t) / it

sizes = [int(a) for a in sys.argv[1:]] or [2048]
net = make_net(dtype=torch.float16, channels_last=True, fold_bn=True)
for bs in sizes:
    s32 = (torch.rand(bs, 18, 19, 19, device="cuda") < 0.3).float()
    s16 = s32.half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = net({"s": s32})
        dt = timeit(lambda: net({"s": s32}))
    print("bs=%d eager(f32 in): %.2f ms  %.0f pos/s" % (bs, dt * 1e3, bs / dt), flush=True)
    for ep in ("hip",):
        try:
            f = FusedInferenceNet(net)
            out = f({"s": s16})
            err = (out["pi"] - ref["pi"]).abs().max().item()
            dt = timeit(lambda: f({"s": s16}))
            print("bs=%d fused[%s](f16 nhwc in): %.2f ms  %.0f pos/s  max|dpi|=%.2e" % (bs, ep, dt * 1e3, bs / dt, err), flush=True)
        except Exception as e:
            print("bs=%d fused[%s] failed: %s" % (bs, ep, repr(e)[:300]), flush=True)
    # conv alone and epilogue alone
    x = torch.randn(bs, 256, 19, 19, device="cuda").half().contiguous(memory_format=torch.channels_last)
    c = net.resnet[0].lower[0]
    with torch.no_grad():
        dt = timeit(lambda: torch.nn.functional.conv2d(x, c.weight, None, 1, 1), it=20)
        print("bs=%d conv3x3 256->256 no bias: %.3f ms  %.0f TFLOP/s" % (bs, dt * 1e3, 2 * 256 * 256 * 9 * 361 * bs / dt / 1e12), flush=True)
        dt = timeit(lambda: torch.nn.functional.conv2d(x, c.weight, c.bias, 1, 1), it=20)
        print("bs=%d conv3x3 256->256 + bias: %.3f ms" % (bs, dt * 1e3), flush=True)
        f = FusedInferenceNet(net)
        y = x.clone()
        dt = timeit(lambda: f._ep(y, c.bias, None), it=20)
        nb = y.numel() * 2
        print("bs=%d epilogue bias+relu: %.3f ms  %.0f GB/s" % (bs, dt * 1e3, 2 * nb / dt / 1e9), flush=True)
        dt = timeit(lambda: f._ep(y, c.bias, x), it=20)
        print("bs=%d epilogue bias+res+relu: %.3f ms  %.0f GB/s" % (bs, dt * 1e3, 3 * nb / dt / 1e9), flush=True)
# feature kernels
B = 16384
eng = elf_amd.GoEngine(19, B, 0)
import numpy as np
seeds = torch.from_numpy((np.arange(B, dtype=np.uint64) * np.uint64(0x9E3779B9) + np.uint64(1)).view(np.int64)).cuda()
eng.playout(seeds, max_steps=120)
d4 = torch.arange(B, device="cuda", dtype=torch.int32) % 8
for fmt, nbytes in (("f32_nchw", 25992 + 736), ("f16_nhwc", 12996 + 736)):
    out = eng.extract_agz(None, d4, n=B, fmt=fmt)
    dt = timeit(lambda: eng.extract_agz(None, d4, out=out, n=B, fmt=fmt), it=20)
    print("extract_agz %s x%d: %.1f us  %.0f GB/s algorithmic" % (fmt, B, dt * 1e6, B * nbytes / dt / 1e9), flush=True)
