"""GPU-box helper: throughput of the random-init 20x256 policy/value net on PyTorch-ROCm."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elf_amd.net import make_net
torch.backends.cudnn.benchmark = True
for dtype in (torch.float16, torch.bfloat16):
    for cl in (True, False):
        net = make_net(dtype=dtype, channels_last=cl)
        for bs in (256, 1024, 2048, 4096):
            s = (torch.rand(bs, 18, 19, 19, device="cuda") < 0.3).float()
            with torch.no_grad():
                for _ in range(3): net({"s": s})
                torch.cuda.synchronize(); t = time.time(); it = 5
                for _ in range(it): net({"s": s})
                torch.cuda.synchronize(); dt = (time.time() - t) / it
            print("%s cl=%d bs=%d: %.2f ms  %.0f pos/s  %.1f TFLOP/s" % (str(dtype).split('.')[-1], cl, bs, dt * 1e3, bs / dt, bs / dt * 17.46e9 / 1e12), flush=True)
