#!/usr/bin/env python
"""GPU-box diagnostic: what in the random-init net's replies makes k_mcts_expand slow?  2048 games, end_step timed with events for several
kinds of reply rows built from the SAME real-net output: as is; values permuted inside each row; ties broken by a tiny ramp; random16."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import elf_amd  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    G = int(os.environ.get("G", "2048"))
    args = argparse.Namespace(net="resnet", net_blocks=20, net_dim=256, net_dtype="fp16", no_fold_bn=False, net_impl="fused", board_size=19)
    net, dtype = bench.build_net(args, 19, dev)
    mk = lambda: elf_amd.SelfPlay(board_size=19, num_games=G, device=0, mcts_rollout_per_thread=2048, mcts_rollout_per_batch=16, mcts_puct=1.5,
                                  mcts_virtual_loss=1, mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, ply_pass_enabled=0,
                                  policy_distri_cutoff=30, seed=1234, nodes_per_game=4096, feature_format="f16_nhwc")
    sp = mk()
    # real replies for the first 6 steps
    real = []
    for _ in range(6):
        sp.begin_step(wait_rows=False)
        with torch.no_grad():
            o = bench.chunked_forward(net, sp.s)
        real.append((o["pi"].float().clone(), o["V"].float().clone()))
        sp.end_step(*real[-1])
    sp.close()
    pi0 = real[3][0]
    print("real pi: dtype", pi0.dtype, "stride", pi0.stride(), "distinct per row (first 4):", [int(torch.unique(r).numel()) for r in pi0[:4]],
          "min %.5f max %.5f" % (float(pi0.min()), float(pi0.max())))
    gen = torch.Generator(device=dev).manual_seed(5)

    def variants(pi, v):
        perm = torch.argsort(torch.rand(pi.shape, device=dev, generator=gen), dim=1)
        ramp = torch.arange(pi.shape[1], device=dev, dtype=torch.float32)[None, :] * 1e-9
        r16 = torch.softmax(0.08 * torch.randn(pi.shape, device=dev, generator=gen), dim=1).half().float()
        sorted_desc = torch.sort(pi, dim=1, descending=True)[0]
        return {"real": pi, "real, values permuted inside each row": torch.gather(pi, 1, perm), "real + 1e-9 * index (no ties)": pi + ramp,
                "random16": r16, "real sorted descending": sorted_desc.contiguous(), "uniform": torch.full_like(pi, 1.0 / pi.shape[1])}

    names = list(variants(*real[0]).keys())
    for name in names:
        sp = mk()
        ts = []
        for i in range(6):
            sp.begin_step(wait_rows=False)
            pi = variants(*real[i])[name]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            sp.end_step(pi, real[i][1])
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        sp.close()
        print("%-45s end_step us: %s" % (name, [int(t) for t in ts]))


if __name__ == "__main__":
    main()
