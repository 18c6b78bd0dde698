"""The policy/value net stays on PyTorch-ROCm (BASELINE.json north_star): this module only restates the
ARCHITECTURE of the reference's Model_PolicyValue (src_py/elfgames/go/df_model3.py:113-313: init conv,
num_block residual blocks of two 3x3 conv+BN, 1x1 policy head -> Linear(2*N*N, N*N+1) -> softmax, 1x1 value
head -> Linear(N*N, 256) -> Linear(256, 1) -> tanh) so that benchmarks can run a random-init 20-block/256-channel
net of the right shape and cost.  It is called through the batch interface: forward({"s": ...}) -> {"pi", "V"}.
"""
import torch
import torch.nn as nn


def _conv_bn(cin, cout, k, relu=True):
    layers = [nn.Conv2d(cin, cout, k, padding=k // 2), nn.BatchNorm2d(cout)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class ResBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.lower = _conv_bn(dim, dim, 3)
        self.upper = _conv_bn(dim, dim, 3, relu=False)

    def forward(self, s):
        return torch.relu(self.upper(self.lower(s)) + s)


class PolicyValueNet(nn.Module):
    def __init__(self, board_size=19, num_planes=18, num_block=20, dim=256):
        super().__init__()
        d = board_size * board_size
        self.d = d
        self.init_conv = _conv_bn(num_planes, dim, 3)
        self.resnet = nn.Sequential(*[ResBlock(dim) for _ in range(num_block)])
        self.pi_final_conv = _conv_bn(dim, 2, 1)
        self.value_final_conv = _conv_bn(dim, 1, 1)
        self.pi_linear = nn.Linear(2 * d, d + 1)
        self.value_linear1 = nn.Linear(d, 256)
        self.value_linear2 = nn.Linear(256, 1)

    def forward(self, batch):
        s = batch["s"] if isinstance(batch, dict) else batch
        p = next(self.parameters())
        s = s.to(dtype=p.dtype)
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            s = s.contiguous(memory_format=torch.channels_last)
        s = self.resnet(self.init_conv(s))
        pi = self.pi_linear(self.pi_final_conv(s).reshape(-1, 2 * self.d))
        pi = torch.softmax(pi.float(), dim=1)
        v = torch.relu(self.value_linear1(self.value_final_conv(s).reshape(-1, self.d)))
        v = torch.tanh(self.value_linear2(v)).float().reshape(-1)
        return dict(pi=pi, V=v)


def map_reference_state_dict(sd):
    """Keys of a reference Model_PolicyValue checkpoint (src_py/elfgames/go/df_model3.py:113-313, saved by
    rlpytorch/model_base.py:83-109 as {"state_dict", "step", "options"}) -> keys of PolicyValueNet.  Same tensors, other names:
        [init_conv|pi_final_conv|value_final_conv](.module)?.{0,1}.*   -> unchanged (DataParallel's ".module" dropped)
        resnet(.module)?.resnet.{i}.conv_lower.{0,1}.*                  -> resnet.{i}.lower.{0,1}.*
        resnet(.module)?.resnet.{i}.conv_upper.{0,1}.*                  -> resnet.{i}.upper.{0,1}.*
        pi_linear.* / value_linear1.* / value_linear2.*                 -> unchanged
    Raises KeyError on a key it does not know (nothing is dropped silently)."""
    import re
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd else sd
    out = {}
    for k, v in sd.items():
        k2 = k.replace(".module.", ".")
        m = re.match(r"^resnet\.resnet\.(\d+)\.conv_(lower|upper)\.(.+)$", k2)
        if m:
            out["resnet.%s.%s.%s" % (m.group(1), m.group(2), m.group(3))] = v
        elif re.match(r"^(init_conv|pi_final_conv|value_final_conv)\.[01]\.", k2) or re.match(r"^(pi_linear|value_linear1|value_linear2)\.", k2):
            out[k2] = v
        else:
            raise KeyError("unknown key in a Model_PolicyValue state_dict: " + k)
    return out


def load_reference_checkpoint(path_or_state, board_size=19, device="cpu"):
    """A PolicyValueNet (eval mode) with the weights of a reference checkpoint (save-*.bin) or of its state_dict.  Block count and
    width are read from the checkpoint itself."""
    sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
    sd = map_reference_state_dict(sd)
    blocks = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("resnet."))
    dim = sd["init_conv.0.weight"].shape[0]
    planes = sd["init_conv.0.weight"].shape[1]
    net = PolicyValueNet(board_size, planes, blocks, dim)
    net.load_state_dict(sd, strict=True)
    return net.eval().to(device)


def fold_batchnorm(net):
    """Inference-time algebra, not a different net: every Conv2d+BatchNorm2d(eval) pair becomes one Conv2d with
    w' = w * gamma / sqrt(var + eps), b' = (b - mean) * gamma / sqrt(var + eps) + beta (torch.nn.utils.fusion)."""
    from torch.nn.utils.fusion import fuse_conv_bn_eval
    for mod in net.modules():
        if isinstance(mod, nn.Sequential) and len(mod) >= 2 and isinstance(mod[0], nn.Conv2d) and isinstance(mod[1], nn.BatchNorm2d):
            mod[0] = fuse_conv_bn_eval(mod[0], mod[1])
            mod[1] = nn.Identity()
    return net


def chunked_forward(net, s, chunk_rows=2048):
    """net on the rows of `s` in slices of at most chunk_rows (the last one may be shorter): every convolution then has the shape of a
    chunk_rows-row call whatever the number of games in flight -- one MIOpen find (whose verification pass scales with the batch: tens
    of seconds for a 16 384-row shape on a cold database) instead of one per batch size, less activation memory, and the 2048-row
    call is the fastest per position on this part.  -> dict(pi [rows, A] f32, V [rows] f32)"""
    rows = s.shape[0]
    if rows <= chunk_rows:
        return net({"s": s})
    pi = v = None
    for c0 in range(0, rows, chunk_rows):
        o = net({"s": s[c0:c0 + chunk_rows]})
        if pi is None:
            pi = torch.empty((rows, o["pi"].shape[1]), dtype=o["pi"].dtype, device=s.device)
            v = torch.empty((rows,), dtype=o["V"].dtype, device=s.device)
        pi[c0:c0 + chunk_rows].copy_(o["pi"])
        v[c0:c0 + chunk_rows].copy_(o["V"])
    return dict(pi=pi, V=v)


class GraphedNet:
    """The same forward captured once into a HIP graph for a fixed batch shape (PyTorch's CUDAGraph on ROCm): one graph
    launch per batch instead of ~200 eager kernel launches.  forward(batch) copies nothing: `s_static` IS the tensor
    the search writes leaf features into.  Batches above chunk_rows run as slices inside the one graph (chunked_forward)."""

    def __init__(self, net, s_static, chunk_rows=2048):
        self.net, self.s = net, s_static
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                chunked_forward(net, s_static, chunk_rows)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = chunked_forward(net, s_static, chunk_rows)

    def __call__(self, batch=None):
        self.graph.replay()
        return self.out


def make_net(board_size=19, num_block=20, dim=256, device="cuda", dtype=torch.float16, channels_last=True, seed=0, fold_bn=False):
    """Random-init net (torch.manual_seed(seed)), eval mode, as the benchmark's stand-in for a trained model."""
    torch.manual_seed(seed)
    net = PolicyValueNet(board_size, 18, num_block, dim).eval()
    if fold_bn:
        net = fold_batchnorm(net)
    net = net.to(device=device, dtype=dtype)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    for p in net.parameters():
        p.requires_grad_(False)
    return net


class FusedInferenceNet:
    """Inference-only execution of a BN-folded fp16 channels_last PolicyValueNet with ONE epilogue pass per convolution.

    The convolutions stay PyTorch-ROCm ops (MIOpen / CK implicit GEMM); what changes is what runs between them.  Eager PyTorch
    issues conv -> bias add -> ReLU (-> residual add -> ReLU) as separate elementwise kernels, i.e. five HBM round trips of the
    [B,256,N,N] activation per residual block; here each conv is bias-free and is followed by one in-place pass
    (`elfnet_bias_act_f16`, elf_amd/csrc/net_epilogue.hip): relu(x + b) after the lower conv, relu(x + b + skip) after the upper.
    (PyTorch's own fused MIOpen ops, miopen_convolution_relu / miopen_convolution_add_relu, were measured and rejected: for
    fp16 channels_last MIOpen's fusion plan falls back to naive kernels, > 10x slower.)
    Same function as PolicyValueNet.forward (src_py/elfgames/go/df_model3.py:62-110,224-313) up to fp16 rounding: the fused
    epilogue rounds once where the eager sequence rounds after every kernel."""

    def __init__(self, net):
        import ctypes as C
        from . import _lib
        p = next(net.parameters())
        if p.dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("FusedInferenceNet needs an fp16 or bf16 net")
        self.dtype = p.dtype
        if any(isinstance(m, nn.BatchNorm2d) for m in net.modules()):
            raise ValueError("fold BatchNorm first (make_net(fold_bn=True))")
        self.net, self.C = net, C
        self.L = _lib.lib()   # raises if libelf_amd.so is missing: no silent fallback
        self.check = _lib.check
        conv = lambda seq: seq[0]
        self.first = conv(net.init_conv)
        self.blocks = [(conv(b.lower), conv(b.upper)) for b in net.resnet]

    def _ep(self, x, bias, res, relu=True):
        rows = x.numel() // x.shape[1]
        C = self.C
        fn = self.L.elfnet_bias_act_f16 if self.dtype == torch.float16 else self.L.elfnet_bias_act_bf16
        self.check(fn(C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()),
                                             C.c_void_p(res.data_ptr()) if res is not None else None, rows, x.shape[1], int(relu),
                                             C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return x

    def _conv(self, x, c, res=None):
        y = torch.nn.functional.conv2d(x, c.weight, None, c.stride, c.padding)
        assert y.is_contiguous(memory_format=torch.channels_last)
        return self._ep(y, c.bias, res)

    @torch.no_grad()
    def __call__(self, batch):
        net = self.net
        s = batch["s"] if isinstance(batch, dict) else batch
        if s.dtype != self.dtype:
            s = s.to(self.dtype)
        s = s.contiguous(memory_format=torch.channels_last)   # no-op for SelfPlay(feature_format="f16_nhwc")
        h = self._conv(s, self.first)
        for lo, up in self.blocks:
            h = self._conv(self._conv(h, lo), up, res=h)
        pi = net.pi_linear(net.pi_final_conv(h).reshape(-1, 2 * net.d))
        pi = torch.softmax(pi.float(), dim=1)
        v = torch.relu(net.value_linear1(net.value_final_conv(h).reshape(-1, net.d)))
        v = torch.tanh(net.value_linear2(v)).float().reshape(-1)
        return dict(pi=pi, V=v)
