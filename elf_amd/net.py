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


def fold_batchnorm(net):
    """Inference-time algebra, not a different net: every Conv2d+BatchNorm2d(eval) pair becomes one Conv2d with
    w' = w * gamma / sqrt(var + eps), b' = (b - mean) * gamma / sqrt(var + eps) + beta (torch.nn.utils.fusion)."""
    from torch.nn.utils.fusion import fuse_conv_bn_eval
    for mod in net.modules():
        if isinstance(mod, nn.Sequential) and len(mod) >= 2 and isinstance(mod[0], nn.Conv2d) and isinstance(mod[1], nn.BatchNorm2d):
            mod[0] = fuse_conv_bn_eval(mod[0], mod[1])
            mod[1] = nn.Identity()
    return net


class GraphedNet:
    """The same forward captured once into a HIP graph for a fixed batch shape (PyTorch's CUDAGraph on ROCm): one graph
    launch per batch instead of ~200 eager kernel launches.  forward(batch) copies nothing: `s_static` IS the tensor
    the search writes leaf features into."""

    def __init__(self, net, s_static):
        self.net, self.s = net, s_static
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                net({"s": s_static})
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = net({"s": s_static})

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
