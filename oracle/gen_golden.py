"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libelfref*.so, built in place from
/root/reference by oracle/Makefile).  Run in the build container only:  python oracle/gen_golden.py
The GPU box has no /root/reference; tests read the committed .npz files.

Fixtures
  sgf_406844.npz   BASELINE config 1: ladder_suite/ladder/406844.sgf, per ply: Zobrist hash, info
                   record, legal mask (bit-packed), Tromp-Taylor value; AGZ planes for all 8 D4 codes
                   (bit-packed) on EVERY ply (201 x 8 rows), plus the SHA-256 of each fp32 row (`feat_sha`, what a
                   consumer that only sees the batch tensor can check).
  ladder_suite.npz every SGF in ladder_suite/ladder that replays legally: move list, final hash/ply,
                   final legal mask, final AGZ planes (code 3).
  playout_19.npz / playout_9.npz   config 2/5 protocol: seeds -> (final hash, ply, steps) per board.
  gtest_9x9.npz    positions from base/test/*.cc known-answer tests are ported by hand in
                   tests/test_reference_known_answers.py (not generated here).
"""
import glob
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from pyoracle import Ref, playout_seeds  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
SUITE = "/root/reference/ladder_suite/ladder"


def pack(a):
    return np.packbits(np.asarray(a, dtype=np.uint8).reshape(-1))


def replay(R, moves, players=None):
    s = R.new()
    for i, c in enumerate(moves):
        if players is not None and players[i] != R.info(s)[1]:
            R.free(s)
            return None
        if R.forward(s, c) != 1:
            R.free(s)
            return None
    return s


def main():
    os.makedirs(OUT, exist_ok=True)
    R = Ref(19)
    # ---- config 1
    mv, pl = R.sgf_moves(os.path.join(SUITE, "406844.sgf"))
    s = R.new()
    hashes, infos, masks, vals, feat_ply, feats, shas = [], [], [], [], [], [], []
    for i in range(len(mv) + 1):
        hashes.append(R.hash(s)); infos.append(R.info(s)); masks.append(pack(R.legal_mask(s)))
        vals.append(R.evaluate(s, 7.5))
        feat_ply.append(i)
        rows = [R.extract_agz(s, d) for d in range(8)]
        feats.append(np.stack([pack(r) for r in rows]))
        shas.append(np.stack([np.frombuffer(hashlib.sha256(np.ascontiguousarray(r, np.float32).tobytes()).digest(), np.uint8) for r in rows]))
        if i < len(mv):
            assert pl[i] == R.info(s)[1] and R.forward(s, mv[i]) == 1
    np.savez_compressed(os.path.join(OUT, "sgf_406844.npz"), moves=mv.astype(np.int16), hash=np.array(hashes, np.uint64),
                        info=np.stack(infos).astype(np.int32), mask=np.stack(masks), value=np.array(vals, np.float32),
                        feat_ply=np.array(feat_ply, np.int32), feat=np.stack(feats), feat_sha=np.stack(shas))
    print("sgf_406844: %d moves, final hash %016x" % (len(mv), hashes[-1]))
    # ---- ladder suite
    names, allmv, offs, fh, fp, fm, ff = [], [], [0], [], [], [], []
    for path in sorted(glob.glob(os.path.join(SUITE, "*.sgf"))):
        try:
            m, p = R.sgf_moves(path)
        except IOError:
            continue
        st = replay(R, m, p)
        if st is None:
            continue
        names.append(os.path.basename(path)); allmv.append(m.astype(np.int16)); offs.append(offs[-1] + len(m))
        fh.append(R.hash(st)); fp.append(R.info(st)[0]); fm.append(pack(R.legal_mask(st))); ff.append(pack(R.extract_agz(st, 3)))
        R.free(st)
    np.savez_compressed(os.path.join(OUT, "ladder_suite.npz"), names=np.array(names), moves=np.concatenate(allmv),
                        offsets=np.array(offs, np.int32), final_hash=np.array(fh, np.uint64), final_ply=np.array(fp, np.int32),
                        final_mask=np.stack(fm), final_feat=np.stack(ff))
    print("ladder_suite: %d SGFs replay legally" % len(names))
    # ---- config 2 / 5 protocol
    for n, boards in ((19, 256), (9, 1024)):
        Rn = Ref(n)
        seeds = playout_seeds(boards)
        tot, out = Rn.playout(seeds, threads=8)
        np.savez_compressed(os.path.join(OUT, "playout_%d.npz" % n), seeds=seeds, out=out)
        print("playout_%d: %d boards, %d steps, mean ply %.1f" % (n, boards, tot, out[:, 2].mean()))


if __name__ == "__main__":
    main()
