"""GPU-box helper: throughput of the random-init 20x256 policy/value net on PyTorch-ROCm."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elf_amd.net import make_net, GraphedNet
torch.backends.cudnn.benchmark = True
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def timeit(f, it=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.time() - t) / it
for dtype in (torch.float16, torch.bfloat16):
    for fold in (False, True):
        net = make_net(dtype=dtype, channels_last=True, fold_bn=fold)
        s = (torch.rand(bs, 18, 19, 19, device="cuda") < 0.3).float()
        with torch.no_grad():
            dt = timeit(lambda: net({"s": s}))
        print("%s fold=%d bs=%d eager: %.2f ms  %.0f pos/s" % (str(dtype).split('.')[-1], fold, bs, dt * 1e3, bs / dt), flush=True)
        try:
            gn = GraphedNet(net, s)
            dt = timeit(lambda: gn())
            print("%s fold=%d bs=%d graph: %.2f ms  %.0f pos/s" % (str(dtype).split('.')[-1], fold, bs, dt * 1e3, bs / dt), flush=True)
        except Exception as e:
            print("graph failed:", repr(e)[:300])
