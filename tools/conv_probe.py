"""GPU-box helper: which layout / dtype / batch gives PyTorch-ROCm's 3x3 256->256 convolution (the net's hot op) its best rate."""
import sys, time, torch
torch.backends.cudnn.benchmark = True
def timeit(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.time() - t) / it
w0 = torch.randn(256, 256, 3, 3, device="cuda")
for bs in (2048, 4096):
    for dtype in (torch.float16, torch.bfloat16):
        for cl in (True, False):
            x = torch.randn(bs, 256, 19, 19, device="cuda").to(dtype)
            w = w0.to(dtype)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
            t0 = time.time()
            with torch.no_grad():
                dt = timeit(lambda: torch.nn.functional.conv2d(x, w, None, 1, 1))
            print("bs=%d %s %s: %.3f ms  %.0f TFLOP/s  (find+time %.1f s)" % (bs, str(dtype).split(".")[-1], "NHWC" if cl else "NCHW", dt * 1e3,
                  2 * 256 * 256 * 9 * 361 * bs / dt / 1e12, time.time() - t0), flush=True)
