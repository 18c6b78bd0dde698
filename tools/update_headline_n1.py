#!/usr/bin/env python
"""Refresh profiles/headline_n1.json (the N = 1 values `bench.py --gpus N` lines relate themselves to) from a 1-GPU default bench line.
usage: python tools/update_headline_n1.py gpurun_out/<tag>/bench.json <tag>"""
import json, os, sys
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(src))
assert d["n_gpus"] == 1
out = {"source": "profiles/%s_bench_n1.json (python bench.py, N = 1, default configuration, one MI355X)" % tag,
       "mcts_rollouts_per_sec": d["value"],
       # what scaling_report("selfplay_games_per_sec") relates to: the shortened configuration played end to end (measured games/s)
       "selfplay_games_per_sec": ((d.get("selfplay_games") or {}).get("shortened_run") or {}).get("games_per_sec") or (d.get("selfplay_games") or {}).get("value")}
json.dump(out, open(os.path.join(root, "profiles", "headline_n1.json"), "w"), indent=1)
print(out)
