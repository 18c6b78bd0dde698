"""GPU: glue around the PyTorch-ROCm net -- the fused conv epilogue kernel (elfnet_bias_act_f16) against plain PyTorch, the
fused inference path against the eager net, and fp16 channels_last leaf features feeding it."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def elf(built):
    import elf_amd
    return elf_amd


@pytest.mark.parametrize("rows,ch", [(1, 8), (361, 256), (5000, 64), (19 * 19 * 37, 256)])
@pytest.mark.parametrize("use_bias,use_res,relu", [(True, False, True), (True, True, True), (False, True, False), (False, False, True)])
def test_bias_act_epilogue(elf, rows, ch, use_bias, use_res, relu):
    """x <- act(x + bias + res), fp32 arithmetic and ONE rounding to fp16 -- compared with the same formula in torch fp32."""
    import torch
    L = elf.lib()
    g = torch.Generator(device="cuda").manual_seed(rows * 131 + ch)
    x = torch.randn((rows, ch), device="cuda", generator=g).half()
    b = torch.randn((ch,), device="cuda", generator=g).half() if use_bias else None
    r = torch.randn((rows, ch), device="cuda", generator=g).half() if use_res else None
    want = x.float()
    if use_bias:
        want = want + b.float()
    if use_res:
        want = want + r.float()
    if relu:
        want = torch.relu(want)
    want = want.half()
    y = x.clone()
    rc = L.elfnet_bias_act_f16(C.c_void_p(y.data_ptr()), C.c_void_p(b.data_ptr()) if use_bias else None,
                               C.c_void_p(r.data_ptr()) if use_res else None, rows, ch, int(relu),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    # x + bias + res in fp32 is exact up to the association of three terms; tolerance = 1 fp16 ulp of the result
    assert torch.allclose(y.float(), want.float(), rtol=2 ** -10, atol=1e-4)
    if not (use_bias and use_res):
        assert torch.equal(y, want)   # at most two addends: bit-exact


@pytest.mark.parametrize("use_res", [False, True])
def test_bias_act_epilogue_bf16(elf, use_res):
    """the bfloat16 variant: fp32 arithmetic, one round-to-nearest-even to bf16"""
    import torch
    L = elf.lib()
    rows, ch = 3000, 256
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((rows, ch), device="cuda", generator=g).bfloat16()
    b = torch.randn((ch,), device="cuda", generator=g).bfloat16()
    r = torch.randn((rows, ch), device="cuda", generator=g).bfloat16() if use_res else None
    want = x.float() + b.float()
    if use_res:
        want = want + r.float()
    want = torch.relu(want).bfloat16()
    y = x.clone()
    assert L.elfnet_bias_act_bf16(C.c_void_p(y.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(r.data_ptr()) if use_res else None,
                                  rows, ch, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    if use_res:
        assert torch.allclose(y.float(), want.float(), rtol=2 ** -7, atol=1e-3)
    else:
        assert torch.equal(y, want)


def test_bias_act_argument_errors(elf):
    import torch
    L = elf.lib()
    x = torch.zeros((4, 16), device="cuda", dtype=torch.float16)
    assert L.elfnet_bias_act_f16(None, None, None, 4, 16, 1, None) < 0
    assert L.elfnet_bias_act_f16(C.c_void_p(x.data_ptr()), None, None, 4, 12, 1, None) < 0        # channels % 8
    assert L.elfnet_bias_act_f16(C.c_void_p(x.data_ptr() + 2), None, None, 1, 8, 1, None) < 0    # alignment
    assert L.elfnet_bias_act_f16(C.c_void_p(x.data_ptr()), None, None, 0, 16, 1, None) == 0


def test_fused_inference_matches_eager_net(elf):
    """Same function as the eager BN-folded net up to fp16 rounding (tolerance: 2e-3 absolute on pi and V, measured against
    an fp32 evaluation of the same weights: the fused path must be at least as close to fp32 as eager fp16 + 1e-3)."""
    import torch
    from elf_amd.net import FusedInferenceNet, make_net
    n, blocks, dim, bs = 19, 4, 64, 48
    net16 = make_net(n, blocks, dim, "cuda", torch.float16, channels_last=True, seed=3, fold_bn=True)
    net32 = make_net(n, blocks, dim, "cuda", torch.float32, channels_last=False, seed=3, fold_bn=True)
    s = (torch.rand((bs, 18, n, n), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) < 0.3).float()
    with torch.no_grad():
        ref = net32({"s": s})
        eager = net16({"s": s})
    fused = FusedInferenceNet(net16)({"s": s})
    for k in ("pi", "V"):
        err_f = (fused[k] - ref[k]).abs().max().item()
        err_e = (eager[k] - ref[k]).abs().max().item()
        assert err_f <= err_e + 1e-3 and err_f < 2e-3, (k, err_f, err_e)
    assert torch.allclose(fused["pi"].sum(1), torch.ones(bs, device="cuda"), atol=1e-4)


def test_selfplay_f16_nhwc_features_equal_f32(elf):
    """SelfPlay(feature_format="f16_nhwc") writes the same leaf rows as the fp32 reference layout, step for step, and the
    search statistics under the same replies are identical."""
    import torch
    kw = dict(board_size=9, num_games=6, mcts_rollout_per_thread=64, mcts_rollout_per_batch=8, mcts_puct=1.5, mcts_virtual_loss=1,
              mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, seed=77, nodes_per_game=1024, log_searches=12)
    a = elf.SelfPlay(feature_format="f32_nchw", **kw)
    b = elf.SelfPlay(feature_format="f16_nhwc", **kw)
    assert b.s.dtype == torch.float16 and b.s.is_contiguous(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda").manual_seed(1)
    for step in range(40):
        ra, rb = a.begin_step(), b.begin_step()
        assert ra == rb
        torch.cuda.synchronize()
        assert torch.equal(a.s[:ra], b.s[:rb].float()), step
        pi = torch.softmax(3.0 * torch.randn((a.max_rows, 82), device="cuda", generator=g), dim=1)
        v = torch.round(torch.tanh(torch.randn((a.max_rows,), device="cuda", generator=g)) * 256) / 256
        a.end_step(pi, v)
        b.end_step(pi, v)
    la, lb = a.search_log(), b.search_log()
    assert len(la[0]) == len(lb[0]) > 0
    for x, y in zip(la[1:], lb[1:]):
        assert np.array_equal(x, y)
    a.close(); b.close()
