"""GPU: device-resident MCTS + self-play host loop vs golden fixtures produced by the REAL reference stack
(oracle/gen_golden_mcts.py: elf::Context + GoGameSelfPlay + MCTSGoAI + tree_search/*.h), same stub net.
Bar: bit-exact -- root edge ORDER (unordered_map iteration order), priors (after Dirichlet), visit counts,
accumulated rewards, most-visited action, sampled move, root value, for every search of the fixture."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from pyoracle import stub_net

pytestmark = pytest.mark.gpu


def run_case(elf_amd, name, max_searches=None):
    import torch
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    m = len(g["move_played"]) if max_searches is None else min(max_searches, len(g["move_played"]))
    sp = elf_amd.SelfPlay(
        board_size=n, num_games=1, device=0, mcts_rollout_per_thread=int(cfg["rollouts_per_thread"]),
        mcts_rollout_per_batch=int(cfg["rollouts_per_batch"]), mcts_puct=float(np.float32(cfg["c_puct"])),
        mcts_virtual_loss=int(cfg["virtual_loss"]), mcts_use_prior=bool(cfg["use_prior"]),
        mcts_persistent_tree=bool(cfg["persistent_tree"]), mcts_epsilon=float(np.float32(cfg["root_epsilon"])),
        mcts_alpha=float(np.float32(cfg["root_alpha"])), mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]),
        mcts_root_unexplored_q_zero=bool(cfg["root_unexplored_q_zero"]), komi=float(np.float32(cfg["komi"])),
        ply_pass_enabled=int(cfg["ply_pass_enabled"]), policy_distri_cutoff=int(cfg["policy_distri_cutoff"]),
        move_cutoff=int(cfg["move_cutoff"]), resign_thres=float(np.float32(cfg["resign_thres"])),
        never_resign_prob=float(np.float32(cfg["never_resign_prob"])), seed=int(cfg["seed"]), log_searches=m)
    salt, ties = int(cfg["net_salt"]), int(cfg["net_tie_levels"])
    rows_total = 0
    while sp.stats()["logged"] < m:
        rows = sp.begin_step()
        rows_total += rows
        if rows:
            pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), salt, ties)
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        else:
            sp.end_step(None, None)
    rec, coord, visits, prior, reward = sp.search_log()
    na = n * n + 1
    for i in range(m):
        ne = int(g["n_edges"][i])
        ctx = "%s search %d" % (name, i)
        assert rec[i].n_edges == ne, ctx
        assert np.array_equal(coord[i, :ne], g["coord"][i, :ne].astype(np.int32)), ctx + ": edge iteration order"
        assert np.array_equal(prior[i, :ne].view(np.uint32), g["prior"][i, :ne].view(np.uint32)), ctx + ": priors"
        assert np.array_equal(visits[i, :ne], g["visits"][i, :ne]), ctx + ": visit counts"
        assert np.array_equal(reward[i, :ne].view(np.uint32), g["reward"][i, :ne].view(np.uint32)), ctx + ": rewards"
        assert rec[i].best_action == int(g["best_action"][i]), ctx
        assert rec[i].total_visits == int(g["total_visits"][i]), ctx
        assert np.float32(rec[i].root_value) == g["root_value"][i], ctx
        assert rec[i].move_played == int(g["move_played"][i]), ctx + ": move played"
    assert na <= coord.shape[1]
    sp.close()
    return rows_total


@pytest.fixture(scope="module")
def elf(built):
    import elf_amd
    return elf_amd


@pytest.mark.parametrize("name", ["mcts_19_r128_fresh", "mcts_19_r256_dir", "mcts_19_r256_ties", "mcts_19_r512_client",
                                  "mcts_9_r512", "mcts_9_r64_ties", "mcts_19_r128_vl0", "mcts_19_r128_noprior", "mcts_9_r128_rootq0",
                                  "mcts_9_r96_bs4", "mcts_9_r128_bs64"])
def test_search_matches_reference(elf, name):
    run_case(elf, name)


def test_config3_8192_rollouts(elf):
    """BASELINE config 3 search settings: bs 16, 8192 rollouts/move, puct 1.5, vloss 1, eps 0.25 / alpha 0.03."""
    run_case(elf, "mcts_19_r8192")


def test_exhausted_node_pool_is_a_status_code(elf):
    """A tree that outgrows nodes_per_game surfaces as ELFGO_E_MCTS_BASE - ELFMCTS_E_POOL (an exception in Python), not as
    corrupted statistics or a hang."""
    import torch
    sp = elf.SelfPlay(board_size=9, num_games=2, mcts_rollout_per_thread=512, mcts_rollout_per_batch=16, nodes_per_game=64, seed=3)
    g = torch.Generator(device="cuda").manual_seed(0)
    with pytest.raises(elf.ElfGoError) as e:
        for _ in range(64):
            rows = sp.begin_step()
            pi = torch.softmax(torch.randn((sp.max_rows, 82), device="cuda", generator=g), dim=1)
            v = torch.zeros(sp.max_rows, device="cuda")
            sp.end_step(pi, v)
    assert "-101" in str(e.value)
    sp.close()


LIVE = [
    (9, dict(rollouts_per_thread=96, max_searches=40, seed=4242, net_salt=77, policy_distri_cutoff=9, virtual_loss=2, c_puct=1.1,
             root_epsilon=0.3, root_alpha=0.2, ply_pass_enabled=12, komi=6.5)),
    (9, dict(rollouts_per_thread=48, rollouts_per_batch=12, batchsize=12, max_searches=60, seed=31337, net_salt=78, net_tie_levels=6,
             persistent_tree=0, unexplored_q_zero=1, move_cutoff=25)),
    (19, dict(rollouts_per_thread=80, max_searches=7, seed=2718, net_salt=79, policy_distri_cutoff=3, virtual_loss=4, c_puct=2.0)),
]


@pytest.mark.parametrize("case", range(len(LIVE)))
def test_live_differential_against_the_reference_stack(elf, case):
    """Not a committed fixture: the REAL reference self-play stack (oracle/_ref/libelfsp*.so, prebuilt, travels with the repo)
    is run here on the host cores with a configuration no fixture uses, and the GPU engine must reproduce every search of it.
    Where the prebuilt reference is absent (a fresh clone), its CPU restatement (oracle/mcts_oracle.cc, pinned on the reference's
    fixtures) takes its place, so the test never skips."""
    import torch
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay, RefSelfPlay
    n, kw = LIVE[case]
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(kw)
    ref = (RefSelfPlay(n) if RefSelfPlay.available(n) else PortSelfPlay(n)).run(**cfg)
    S = ref["search"]
    m = len(S)
    assert m == cfg["max_searches"]
    sp = elf.SelfPlay(
        board_size=n, num_games=1, device=0, mcts_rollout_per_thread=cfg["rollouts_per_thread"], mcts_rollout_per_batch=cfg["rollouts_per_batch"],
        mcts_puct=cfg["c_puct"], mcts_virtual_loss=cfg["virtual_loss"], mcts_use_prior=bool(cfg["use_prior"]),
        mcts_persistent_tree=bool(cfg["persistent_tree"]), mcts_epsilon=cfg["root_epsilon"], mcts_alpha=cfg["root_alpha"],
        mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]), mcts_root_unexplored_q_zero=bool(cfg["root_unexplored_q_zero"]),
        komi=cfg["komi"], ply_pass_enabled=cfg["ply_pass_enabled"], policy_distri_cutoff=cfg["policy_distri_cutoff"],
        move_cutoff=cfg["move_cutoff"], resign_thres=cfg["resign_thres"], never_resign_prob=cfg["never_resign_prob"], seed=cfg["seed"],
        log_searches=m, nodes_per_game=4096)
    while sp.stats()["logged"] < m:
        rows = sp.begin_step()
        if rows:
            pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), cfg["net_salt"], cfg["net_tie_levels"])
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        else:
            sp.end_step(None, None)
    rec, coord, visits, prior, reward = sp.search_log()
    for i in range(m):
        ne = S[i].n_edges
        ctx = "live case %d search %d" % (case, i)
        assert rec[i].n_edges == ne, ctx
        assert np.array_equal(coord[i, :ne], ref["coord"][i, :ne]), ctx
        assert np.array_equal(visits[i, :ne], ref["visits"][i, :ne]), ctx
        assert np.array_equal(prior[i, :ne].view(np.uint32), ref["prior"][i, :ne].view(np.uint32)), ctx
        assert np.array_equal(reward[i, :ne].view(np.uint32), ref["reward"][i, :ne].view(np.uint32)), ctx
        assert rec[i].move_played == S[i].move_played and rec[i].best_action == S[i].best_action, ctx
        assert np.float32(rec[i].root_value) == np.float32(S[i].root_value), ctx
    sp.close()
