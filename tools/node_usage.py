#!/usr/bin/env python
"""GPU-box tool: how many node ids does a game's persistent tree really hold at the headline's search settings (8192 rollouts per move,
bs 16)?  Plays `--moves` moves with `--games` games, samples the free-id count of every game's small pool during the searches and prints
the peak number of live nodes per game (max over games and over time) -- what nodes_per_game has to cover.
    python tools/node_usage.py [--net random|resnet] [--games 64] [--moves 4] [--rollouts 8192]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import elf_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="random", help="random: peaky replies without prior ties; random16: near-uniform replies on the fp16 grid (prior ties in every "
                    "row: the expansion's exact std::sort replay); resnet: the benchmark's net")
    ap.add_argument("--games", type=int, default=64)
    ap.add_argument("--moves", type=int, default=4)
    ap.add_argument("--rollouts", type=int, default=8192)
    ap.add_argument("--nodes-per-game", type=int, default=None)
    ap.add_argument("--validate", action="store_true", help="after every move: the node records' invariants (elfmcts_validate) and the pool's "
                    "books (ids in trees by a scan = RootInfo = pool_info; trees + free = total)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(net="random" if a.net == "random16" else a.net, net_blocks=20, net_dim=256, net_dtype="fp16", no_fold_bn=False, net_impl="fused", board_size=19)
    net, dtype = bench.build_net(args, 19, dev)
    sp = elf_amd.SelfPlay(board_size=19, num_games=a.games, device=0, mcts_rollout_per_thread=a.rollouts, mcts_rollout_per_batch=16,
                          mcts_puct=1.5, mcts_virtual_loss=1, mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5,
                          ply_pass_enabled=0, policy_distri_cutoff=30, seed=1234, nodes_per_game=a.nodes_per_game,
                          feature_format="f16_nhwc" if net is not None else "f32_nchw")
    L = elf_amd.lib()
    m = L.elfsp_mcts(sp._h)
    info = torch.zeros((a.games, 8), dtype=torch.int32, device=dev)
    rnd = bench.RandomReplies(sp.max_rows, 362, dev, 7, flat16=(a.net == "random16"))
    cs = int(sp.opt.nodes_per_game)
    peak, per_move = np.zeros(a.games, np.int64), []
    spm = sp.stats()["steps_per_move"]
    for mv in range(a.moves):
        for st in range(spm):
            sp.begin_step(wait_rows=False)
            if net is not None:
                with torch.no_grad():
                    o = net({"s": sp.s})
                pi, v = o["pi"], o["V"]
            else:
                pi, v = rnd()
            if st == spm - 1:      # the tree is largest just before the move is played (treeAdvance frees the siblings' subtrees)
                L.elfmcts_root(m, C.c_void_p(info.data_ptr()), None, None, None, None, None, None)
                torch.cuda.synchronize()
                used = info[:, 7].cpu().numpy().astype(np.int64)      # RootInfo word 7: node ids the game's tree holds
                peak = np.maximum(peak, used)
                per_move.append({"move": mv + 1, "live_nodes_mean": float(used.mean()), "live_nodes_max": int(used.max())})
            sp.end_step(pi, v)
        if a.validate:
            bad = sp.validate_trees()
            p_, live = sp.pool_info(), sp.count_live()
            assert bad[0] == 0, ("node record invariants", bad)
            assert int(live.sum()) == p_["live"] and p_["live"] + p_["small_free"] + p_["big_free"] == p_["small_total"] + p_["big_total"], p_
            per_move[-1].update(validated=True, ids_in_trees=p_["live"], big_records_in_use=p_["big_total"] - p_["big_free"])
    pool = sp.pool_info()
    print(json.dumps({"net": a.net, "games": a.games, "rollouts_per_move": a.rollouts, "nodes_per_game": cs, "per_move": per_move, "pool": pool,
                      "pool_GB": a.games * elf_amd.tree_bytes_per_game(19, cs) / 1e9,
                      "fixed_pools_would_need_ids_per_game": int(peak.max()), "shared_pool_peak_ids_per_game": max(p["live_nodes_mean"] for p in per_move),
                      "peak_live_nodes_over_games": int(peak.max()), "peak_over_rollouts": float(peak.max()) / a.rollouts,
                      "bytes_per_game": elf_amd.tree_bytes_per_game(19, cs)}))
    sp.close()


if __name__ == "__main__":
    main()
