"""Generates tests/golden/wire_formats.json from the REAL reference objects (oracle/_ref/libelfsp9.so: Records, ThreadState,
MsgRequestSeq of src_cpp/elfgames/go/common/record.h): the texts a self-play client exchanges with the training server
(train/distri_client.h).  Run in the build container only:  python oracle/gen_golden_wire.py
  requests: what the server writes for a request (MsgRequestSeq::dumpJsonString) + variants with fields removed and what
            MsgRequestSeq::createFromJson makes of them (round-tripped text, or null = it throws)
  sessions: GuardedRecords sessions -- operations (state updates, finished games, dumps) and the message text of every dump
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from pyoracle import RefSelfPlay  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    R = RefSelfPlay(9)
    requests = []
    cases = [
        dict(black_ver=12, white_ver=-1, num_game_thread_used=-1, black_thres=0.05, white_thres=0.05, never_resign_prob=0.1, seq=7,
             rollouts_per_thread=1600, rollouts_per_batch=8, mcts_threads=2, virtual_loss=5, c_puct=0.85, root_epsilon=0.25, root_alpha=0.03),
        dict(black_ver=40, white_ver=39, client_type=2, num_game_thread_used=8, black_thres=0.01, white_thres=0.02, player_swap=1, seq=123456789012,
             rollouts_per_thread=400, rollouts_per_batch=16, persistent_tree=0, unexplored_q_zero=1, root_unexplored_q_zero=1, use_prior=0),
        dict(black_ver=-1, white_ver=-1, seq=0),                                          # a wait request
        dict(black_ver=5, white_ver=-1, async_=1, client_type=1, pick_method=1, seq=3, c_puct=1.5, root_epsilon=0.0, root_alpha=0.0),
        dict(black_ver=3000000000, white_ver=-1, pick_method=2, seq=-1, never_resign_prob=0.333),
    ]
    for kw in cases:
        text = R.request_seq_dump(**kw)
        assert R.request_seq_roundtrip(text) == text
        variants = []
        j = json.loads(text)
        for path in (("request", "client_ctrl", "async"), ("request", "client_ctrl", "player_swap"), ("request", "client_ctrl", "client_type"),
                     ("request", "vers", "mcts_opt", "alg_opt", "c_puct"), ("request", "vers", "mcts_opt", "pick_method"),
                     ("request", "vers", "white_ver"), ("seq",), ("request", "client_ctrl"), ("request", "vers", "mcts_opt", "log_prefix")):
            jj = json.loads(text)
            d = jj
            for k in path[:-1]:
                d = d[k]
            del d[path[-1]]
            t = json.dumps(jj, separators=(",", ":"))
            variants.append(dict(removed="/".join(path), text=t, roundtrip=R.request_seq_roundtrip(t)))
        # a reply with other spacing and key order, as another JSON writer would produce it
        loose = json.dumps(json.loads(text), indent=1, sort_keys=False)
        variants.append(dict(removed="", text=loose, roundtrip=R.request_seq_roundtrip(loose)))
        requests.append(dict(params=kw, text=text, variants=variants))
    recs = [str(t) for t in np.load(os.path.join(OUT, "records_9_cutoff.npz"))["records"]]
    recs += [str(t) for t in np.load(os.path.join(OUT, "records_9_eval.npz"))["records"]]
    sessions = []
    rng = np.random.default_rng(5)
    for identity, nthreads in (("gpu-box-17_4242", 40), ("a \"quoted\" \\ id\twith\x01ctrl", 3), ("many", 300)):
        R.client_reset(identity)
        ops, dumps = [], []
        seqs = np.ones(nthreads, np.int64)
        for rnd in range(4):
            order = rng.permutation(nthreads) if rnd != 1 else np.arange(nthreads)[::-1]
            upd = list(order) + [int(x) for x in rng.integers(0, nthreads, size=nthreads // 2)]   # some threads report twice
            if rnd == 3:
                upd = []
            for t in upd:
                st = dict(thread_id=int(t), seq=int(seqs[t]), move_idx=int(rng.integers(-1, 300)), black=int(10 + rnd), white=int(-1 if rnd % 2 == 0 else 9))
                R.client_update_state(st["thread_id"], st["seq"], st["move_idx"], st["black"], st["white"])
                ops.append(dict(op="state", **st))
                if rng.random() < 0.2:
                    seqs[t] += 1
            for k in range(0 if rnd == 2 else int(rng.integers(1, 4))):
                ri = int(rng.integers(0, len(recs)))
                R.client_feed(recs[ri])
                ops.append(dict(op="feed", rec=ri))
            text = R.client_dump()
            ops.append(dict(op="dump"))
            dumps.append(text)
            assert R.records_parse(text) is not None
        sessions.append(dict(identity=identity, ops=ops, dumps=dumps))
    with open(os.path.join(OUT, "wire_formats.json"), "w") as fh:
        json.dump(dict(requests=requests, records=recs, sessions=sessions), fh, separators=(",", ":"))
    print("requests", len(requests), "sessions", [(s["identity"][:12], [len(d) for d in s["dumps"]]) for s in sessions])
    for r in requests[:2]:
        print(r["text"][:300])
        print([(v["removed"], v["roundtrip"] is not None) for v in r["variants"]])


if __name__ == "__main__":
    main()
