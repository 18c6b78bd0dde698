"""GPU-box helper: run one MCTS golden case and print the first mismatch in detail."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import elf_amd
from pyoracle import stub_net
name = sys.argv[1]; maxs = int(sys.argv[2]) if len(sys.argv) > 2 else None
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
n = int(g["board_size"]); m = len(g["move_played"]) if maxs is None else min(maxs, len(g["move_played"]))
sp = elf_amd.SelfPlay(board_size=n, num_games=1, mcts_rollout_per_thread=int(cfg["rollouts_per_thread"]),
    mcts_rollout_per_batch=int(cfg["rollouts_per_batch"]), mcts_puct=float(np.float32(cfg["c_puct"])), mcts_virtual_loss=int(cfg["virtual_loss"]),
    mcts_persistent_tree=bool(cfg["persistent_tree"]), mcts_epsilon=float(np.float32(cfg["root_epsilon"])), mcts_alpha=float(np.float32(cfg["root_alpha"])),
    mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]), komi=float(np.float32(cfg["komi"])), ply_pass_enabled=int(cfg["ply_pass_enabled"]),
    policy_distri_cutoff=int(cfg["policy_distri_cutoff"]), seed=int(cfg["seed"]), log_searches=m)
salt, ties = int(cfg["net_salt"]), int(cfg["net_tie_levels"])
while sp.stats()["logged"] < m:
    rows = sp.begin_step()
    if rows:
        pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), salt, ties)
        sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
    else:
        sp.end_step(None, None)
rec, coord, visits, prior, reward = sp.search_log()
print("stats", sp.stats())
for i in range(m):
    ne = int(g["n_edges"][i])
    ok = dict(
        n_edges=rec[i].n_edges == ne,
        order=np.array_equal(coord[i, :ne], g["coord"][i, :ne].astype(np.int32)),
        prior=np.array_equal(prior[i, :ne].view(np.uint32), g["prior"][i, :ne].view(np.uint32)),
        visits=np.array_equal(visits[i, :ne], g["visits"][i, :ne]),
        reward=np.array_equal(reward[i, :ne].view(np.uint32), g["reward"][i, :ne].view(np.uint32)),
        best=rec[i].best_action == int(g["best_action"][i]), move=rec[i].move_played == int(g["move_played"][i]),
        rootv=np.float32(rec[i].root_value) == g["root_value"][i])
    print(i, "ne", rec[i].n_edges, ne, "move", rec[i].move_played, int(g["move_played"][i]), "tv", rec[i].total_visits, int(g["total_visits"][i]), ok)
    if not all(ok.values()):
        print(" mine  coord", coord[i, :12], "\n gold  coord", g["coord"][i, :12])
        print(" mine  prior", prior[i, :6], "\n gold  prior", g["prior"][i, :6])
        bad = np.nonzero(visits[i, :ne] != g["visits"][i, :ne])[0]
        print(" visit diffs at", bad[:10], visits[i, bad[:10]], g["visits"][i, bad[:10]])
        print(" sorted mine", np.sort(visits[i, :ne])[-8:], "gold", np.sort(g["visits"][i, :ne])[-8:])
        break
