"""Test helpers: an elf_amd.SelfPlay context under the configuration of a reference fixture, driven with the fixture's stub nets
(one AI per game through begin_step / end_step, evaluation games with two AIs through begin_step2 / end_step2)."""
import numpy as np

from pyoracle import stub_net

PICK = {0: "most_visited", 1: "strongest_prior", 2: "uniform_random"}


def sp_from_fixture_cfg(elf_amd, n, cfg, **over):
    """elf_amd.SelfPlay under the configuration a reference fixture was generated with (oracle/pyoracle.py MCTS_DEFAULTS keys)"""
    f32 = lambda k: float(np.float32(cfg[k]))
    kw = dict(
        board_size=n, num_games=int(cfg.get("num_games", 1)), device=0, mcts_rollout_per_thread=int(cfg["rollouts_per_thread"]),
        mcts_rollout_per_batch=int(cfg["rollouts_per_batch"]), mcts_puct=f32("c_puct"),
        mcts_virtual_loss=int(cfg["virtual_loss"]), mcts_use_prior=bool(cfg["use_prior"]),
        mcts_persistent_tree=bool(cfg["persistent_tree"]), mcts_epsilon=f32("root_epsilon"),
        mcts_alpha=f32("root_alpha"), mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]),
        mcts_root_unexplored_q_zero=bool(cfg["root_unexplored_q_zero"]), komi=f32("komi"),
        ply_pass_enabled=int(cfg["ply_pass_enabled"]), policy_distri_cutoff=int(cfg["policy_distri_cutoff"]),
        move_cutoff=int(cfg["move_cutoff"]), resign_thres=f32("resign_thres"),
        never_resign_prob=f32("never_resign_prob"), seed=int(cfg["seed"]), mcts_threads=int(cfg.get("mcts_threads", 1)))
    if "white_ver" in cfg:          # round-3 fixtures: evaluation games, pick methods, policy-only play
        kw.update(white_puct=f32("white_puct"), white_mcts_rollout_per_batch=int(cfg["white_rollouts_per_batch"]),
                  white_mcts_rollout_per_thread=int(cfg["white_rollouts_per_thread"]),
                  black_use_policy_network_only=bool(cfg["black_policy_only"]), white_use_policy_network_only=bool(cfg["white_policy_only"]),
                  mcts_pick_method=PICK[int(cfg["pick_method"])], model_ver=int(cfg["black_ver"]))
    if "cheat_selfplay_random_result" in cfg:
        kw.update(cheat_eval_new_model_wins_half=bool(cfg["cheat_eval_new_model_wins_half"]),
                  cheat_selfplay_random_result=bool(cfg["cheat_selfplay_random_result"]))
    kw.update(over)
    sp = elf_amd.SelfPlay(**kw)
    if "white_ver" in cfg and (int(cfg["white_ver"]) >= 0 or int(cfg["black_ver"]) != 0 or int(cfg["thread_used"]) != 0):
        tu = int(cfg["thread_used"]) or kw["num_games"]
        sp.set_request(int(cfg["black_ver"]), int(cfg["white_ver"]), f32("resign_thres"), f32("never_resign_prob"),
                       num_game_thread_used=tu, player_swap=bool(cfg["player_swap"]))
    return sp


def drive_stub(sp, n, cfg, done, on_step=None, black_ver=None):
    """serve batches with the stub nets of the fixture until done(sp); two-AI games through begin_step2 / end_step2 with the
    "actor_white" rows evaluated by the second stub net and every reply carrying its model's version in rv (black_ver: a callable
    giving the (black, white) versions the models have now, for runs in which a later request changes them)"""
    import torch
    salt, ties = int(cfg["net_salt"]), int(cfg["net_tie_levels"])
    two = int(cfg.get("white_ver", -1)) >= 0 or int(cfg.get("req2_white_ver", -1)) >= 0
    bv, wv = int(cfg.get("black_ver", 0)), int(cfg.get("white_ver", -1))
    rows_total = [0, 0]
    while not done(sp):
        if two:
            rb, rw = sp.begin_step2()
            if black_ver is not None:          # a later request changes the models: (black, white) versions now
                bv, wv = black_ver()
            rep = [None, None]
            for a, (rows, s, sl, ver) in enumerate(((rb, sp.s, salt, bv), (rw, sp.s_white, int(cfg["white_net_salt"]), wv))):
                if rows:
                    pi, v = stub_net(n, s[:rows].cpu().numpy(), sl, ties)
                    rep[a] = (torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device),
                              torch.full((rows,), ver, dtype=torch.int64, device=sp.device))
                rows_total[a] += rows
            sp.end_step2(rep)
        else:
            rows = sp.begin_step()
            rows_total[0] += rows
            if rows:
                pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), salt, ties)
                sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device),
                            torch.full((rows,), bv if black_ver is None else black_ver()[0], dtype=torch.int64, device=sp.device))
            else:
                sp.end_step(None, None)
        if on_step:
            on_step(sp, rows_total)
    return rows_total


