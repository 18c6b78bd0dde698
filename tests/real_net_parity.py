"""SURVEY.md 8(d) config 3 parity artefact, as the survey defines it: the REAL reference stack (oracle/_ref/libelfsp19.so:
elf::Context batcher + GoGameSelfPlay + MCTSGoAI + tree_search/*.h compiled in place) and the HIP engine search with the SAME
net -- Model_PolicyValue 20 blocks x 256 channels, torch.manual_seed(0) default init, eval mode, fp32 -- on the same GPU, and
root statistics are compared search by search: edge order, priors, visit counts, rewards, move played.

TEST INFRASTRUCTURE (used by tests/test_gpu_mcts.py, bench.py's parity note and tools/profile_all.sh); the product path does
not import it.

"The same net" is made a pure function of the feature row: rows are evaluated in fixed batches of 16 (one convolution
algorithm for every call) and the result is memoised by a 128-bit digest of the row's bytes, so both engines receive bit-identical
(pi, V) for the same position + D4 code whatever batch the row arrived in.

Hazard H2: the reference backs the leaves of a batch up in the iteration order of an
unordered_map keyed by heap addresses (tree_search.h:216,245); the engine uses first-occurrence order.  With an un-quantised
value head the fp32 reward sums of an edge that received two or more backups from ONE batch can differ in the last ulps, and
a PUCT comparison that is decided by those ulps can then flip.  compare() reports, per game, the first search whose statistics
differ and how (rewards only / visit counts / move)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


class MemoNet:
    """fp32 PolicyValueNet as a pure function of one feature row (see module docstring)."""

    def __init__(self, net, dev, n, pad=16):
        self.net, self.dev, self.n, self.pad = net, dev, n, pad
        self.cache = {}
        self.calls = self.rows = self.misses = 0

    def __call__(self, s):
        import torch
        b = s.shape[0]
        s = np.ascontiguousarray(s, dtype=np.float32)
        keys = [hashlib.blake2b(s[i].tobytes(), digest_size=16).digest() for i in range(b)]
        miss = [i for i in range(b) if keys[i] not in self.cache]
        self.calls += 1
        self.rows += b
        self.misses += len(miss)
        # distinct positions only, in fixed batches of `pad` rows
        seen = {}
        for i in miss:
            seen.setdefault(keys[i], i)
        todo = list(seen.values())
        for c0 in range(0, len(todo), self.pad):
            idx = todo[c0:c0 + self.pad]
            x = np.zeros((self.pad, 18, self.n, self.n), np.float32)
            x[:len(idx)] = s[idx]
            with torch.no_grad():
                out = self.net({"s": torch.from_numpy(x).to(self.dev)})
                pi = out["pi"].float().cpu().numpy()
                v = out["V"].float().cpu().numpy()
            for j, i in enumerate(idx):
                self.cache[keys[i]] = (pi[j].copy(), np.float32(v[j]))
        pi = np.stack([self.cache[k][0] for k in keys])
        v = np.array([self.cache[k][1] for k in keys], np.float32)
        return pi, v


def make_memo_net(n=19, num_block=20, dim=256, device=0):
    import torch
    from elf_amd.net import make_net
    dev = torch.device("cuda", device)
    # torch.manual_seed(0), default init, eval, fp32; deterministic kernels where PyTorch offers the choice
    torch.backends.cudnn.benchmark = False
    net = make_net(n, num_block, dim, device=dev, dtype=torch.float32, channels_last=False, seed=0)
    return MemoNet(net, dev, n)


def search_cfg(**over):
    """SURVEY.md 8(d) config 3 search settings (start_selfplay.sh:33-57 + start_client.sh:15-16 for the Dirichlet noise)"""
    from pyoracle import MCTS_DEFAULTS
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=1, batchsize=16, mcts_threads=1, rollouts_per_batch=16, virtual_loss=1, persistent_tree=1, use_prior=1,
               c_puct=1.5, root_epsilon=0.25, root_alpha=0.03, komi=7.5, ply_pass_enabled=0, seed=1234, policy_distri_cutoff=0)
    cfg.update(over)
    return cfg


def run_reference(memo, n, cfg, games, moves, canonical_backup=False, turnstile=False, preload=None):
    """the reference stack, one run per game (game g seeded seed + g, the rule of include/elf_amd.h) -> {g: [search tuples]}.
    canonical_backup=True: the build whose batch_rollouts backs the unique leaves of a batch up in first-occurrence order instead of
    heap-address order (oracle/Makefile, libelfsp*_h2.so: three lines of a build-time copy of tree_search.h) -- the order SURVEY.md H2
    asks the oracle to pin; the engine must equal THAT build in every bit."""
    from pyoracle import RefSelfPlay
    out = {}
    R = RefSelfPlay(n, canonical_backup=canonical_backup, turnstile=turnstile)   # turnstile: mcts_threads > 1 under the forced schedule
    if preload is not None:   # GameOptions.preload_sgf / preload_sgf_move_to: (Coords of a game, plies forwarded before the first search)
        import tempfile
        fd, path = tempfile.mkstemp(suffix=".sgf")
        with os.fdopen(fd, "w") as fh:
            fh.write("(;GM[1]FF[4]SZ[%d]KM[7.5]" % n + R.coords2sgfstr(preload[0])[1:])
        R.set_preload(path, int(preload[1]))
    try:
        return _run_reference(R, memo, cfg, games, moves)
    finally:
        if preload is not None:
            R.set_preload("", -1)
            os.unlink(path)


def _run_reference(R, memo, cfg, games, moves):
    out = {}
    for g in range(games):
        c = dict(cfg)
        c.update(num_games=1, seed=cfg["seed"] + g, max_searches=moves)
        r = R.run(net=memo, **c)
        out[g] = [_tuple(r["search"][i], r["coord"][i], r["visits"][i], r["prior"][i], r["reward"][i]) for i in range(len(r["search"]))]
    return out


def run_engine(memo, n, cfg, games, moves, nodes_per_game=None, preload=None):
    import torch
    import elf_amd
    sp = elf_amd.SelfPlay(
        board_size=n, num_games=games, device=0, mcts_rollout_per_thread=cfg["rollouts_per_thread"],
        mcts_rollout_per_batch=cfg["rollouts_per_batch"], mcts_puct=cfg["c_puct"], mcts_virtual_loss=cfg["virtual_loss"],
        mcts_use_prior=bool(cfg["use_prior"]), mcts_persistent_tree=bool(cfg["persistent_tree"]), mcts_epsilon=cfg["root_epsilon"],
        mcts_alpha=cfg["root_alpha"], mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]),
        mcts_root_unexplored_q_zero=bool(cfg["root_unexplored_q_zero"]), komi=cfg["komi"], ply_pass_enabled=cfg["ply_pass_enabled"],
        policy_distri_cutoff=cfg["policy_distri_cutoff"], move_cutoff=cfg["move_cutoff"], resign_thres=cfg["resign_thres"],
        never_resign_prob=cfg["never_resign_prob"], seed=cfg["seed"], mcts_threads=cfg["mcts_threads"], log_searches=games * moves,
        nodes_per_game=nodes_per_game or (4 * cfg["rollouts_per_thread"] + 1024))
    if preload is not None:
        sp.preload(preload[0], int(preload[1]))
    misses0 = memo.misses
    while sp.stats()["logged"] < games * moves:
        rows = sp.begin_step()
        if rows:
            pi, v = memo(sp.s[:rows].cpu().numpy())
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        else:
            sp.end_step(None, None)
    rec, coord, visits, prior, reward = sp.search_log()
    sp.close()
    na = n * n + 1
    out = {g: [] for g in range(games)}
    for i, r in enumerate(rec):
        out[r.game].append(_tuple(r, coord[i, :na], visits[i, :na], prior[i, :na], reward[i, :na]))
    return out, memo.misses - misses0


def _tuple(r, coord, visits, prior, reward):
    ne = r.n_edges
    return dict(n_edges=ne, move_played=int(r.move_played), best_action=int(r.best_action), total_visits=int(r.total_visits),
                root_value=np.float32(r.root_value), coord=np.array(coord[:ne], np.int32), visits=np.array(visits[:ne], np.int32),
                prior=np.array(prior[:ne], np.float32), reward=np.array(reward[:ne], np.float32))


def compare(ref, got, games, moves):
    """-> dict: per game the first differing search and its kind; totals.  A game's searches after its first visit-count or move
    difference are not comparable (different trees / positions) and are not counted."""
    per_game, compared, equal, reward_only, diverged = [], 0, 0, 0, 0
    max_ulps = 0
    for g in range(games):
        first = None
        for k in range(min(moves, len(ref[g]), len(got[g]))):
            a, b = ref[g][k], got[g][k]
            compared += 1
            same_shape = a["n_edges"] == b["n_edges"] and np.array_equal(a["coord"], b["coord"])
            same_prior = same_shape and np.array_equal(a["prior"].view(np.uint32), b["prior"].view(np.uint32))
            same_visits = same_shape and np.array_equal(a["visits"], b["visits"])
            same_move = a["move_played"] == b["move_played"] and a["best_action"] == b["best_action"]
            same_reward = same_shape and np.array_equal(a["reward"].view(np.uint32), b["reward"].view(np.uint32))
            same_root = a["root_value"].tobytes() == b["root_value"].tobytes()
            if same_shape and same_prior and same_visits and same_move and same_reward and same_root:
                equal += 1
                continue
            if same_shape and same_prior and same_visits and same_move and same_root:
                reward_only += 1          # the sums differ in their last bits, every decision was the same
                d = np.abs(a["reward"].view(np.int32).astype(np.int64) - b["reward"].view(np.int32).astype(np.int64))
                max_ulps = max(max_ulps, int(d.max()))
                if first is None:
                    first = dict(search=k, kind="reward_ulps", max_ulps=int(d.max()), edges=int((d > 0).sum()))
                continue
            diverged += 1
            kind = ("edge_order" if not same_shape else "prior" if not same_prior else "visits" if not same_visits else
                    "move" if not same_move else "root_value")
            info = dict(search=k, kind=kind)
            if same_shape and not same_visits:
                dv = a["visits"] - b["visits"]
                info.update(edges_differing=int((dv != 0).sum()), max_abs_visit_diff=int(np.abs(dv).max()),
                            total_visits=[a["total_visits"], b["total_visits"]], same_move=bool(same_move))
            if first is None or first["kind"] == "reward_ulps":
                first = info if first is None else dict(info, preceded_by_reward_ulps_at=first["search"])
            break                          # later searches of this game run on different trees
        per_game.append(first)
    return dict(games=games, moves=moves, searches_compared=compared, bit_equal=equal, reward_ulps_only=reward_only,
                decision_diverged=diverged, max_reward_ulps=max_ulps, first_difference_per_game=per_game)


def measure(n=19, games=8, moves=8, rollouts=512, num_block=20, dim=256, seed=1234, memo=None, canonical_backup=False, threads=1):
    from pyoracle import RefSelfPlay
    if not RefSelfPlay.available(n, canonical_backup=canonical_backup, turnstile=threads > 1):
        raise RuntimeError("oracle/_ref/libelfsp%d*.so is not built (make -C oracle ref, needs /root/reference)" % n)
    memo = memo or make_memo_net(n, num_block, dim)
    cfg = search_cfg(rollouts_per_thread=rollouts, seed=seed, mcts_threads=threads)
    ref = run_reference(memo, n, cfg, games, moves, canonical_backup=canonical_backup, turnstile=threads > 1)
    got, engine_misses = run_engine(memo, n, cfg, games, moves)
    res = compare(ref, got, games, moves)
    res.update(reference_build=("canonical backup order" if canonical_backup else "stock (heap-address backup order)") +
               (", turnstile schedule of %d search threads" % threads if threads > 1 else ""), mcts_threads=threads, rollouts=rollouts, net="PolicyValueNet %dx%d fp32 torch.manual_seed(0) eval" % (num_block, dim), seed=seed,
               net_rows=memo.rows, net_distinct_positions=len(memo.cache), engine_rows_the_reference_never_asked=engine_misses)
    return res


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8)
    ap.add_argument("--moves", type=int, default=8)
    ap.add_argument("--rollouts", type=int, default=512)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--board", type=int, default=19)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=1, help="mcts_threads (> 1: against the turnstile build of the reference)")
    ap.add_argument("--canonical", type=int, default=0, help="1: against the canonical-backup-order build of the reference (0 ulps expected)")
    a = ap.parse_args()
    r = measure(a.board, a.games, a.moves, a.rollouts, a.blocks, a.dim, a.seed, canonical_backup=bool(a.canonical), threads=a.threads)
    txt = json.dumps(r, default=lambda o: o.item() if hasattr(o, "item") else str(o))
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
