"""Writes elf_amd/data/zobrist21.bin: the 441 64-bit Zobrist constants of the reference
(src_cpp/elfgames/go/base/hash_num.h:12), read out of the *compiled* reference
(oracle/_ref/libelfref19.so -> ref_zobrist) as little-endian u64, indexed by reference Coord.
The constants are data the parity check needs verbatim; run once in
the build container:  python oracle/gen_zobrist.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pyoracle import Ref, ZOBRIST_BIN  # noqa: E402

z = Ref(19).zobrist()
assert z.size == 441
os.makedirs(os.path.dirname(ZOBRIST_BIN), exist_ok=True)
z.astype("<u8").tofile(ZOBRIST_BIN)
print("wrote", ZOBRIST_BIN, z.size, "constants; first = %016x" % int(z[0]))
