"""elf_amd: MI355X-native (gfx950) implementation of the ELF OpenGo self-play hot path.

The product path is libelf_amd.so (hand-written HIP, elf_amd/csrc) behind the C ABI in
include/elf_amd.h; this package is the Python host side mirroring the reference's interface.
Importing never touches oracle/ and never falls back to a CPU implementation.
"""
from ._lib import ElfGoError, lib  # noqa: F401
from .engine import (  # noqa: F401
    GoEngine, M_PASS, M_RESIGN, M_SKIP, M_INVALID, M_CLEAR, S_EMPTY, S_BLACK, S_WHITE,
    coord, coord_xy, coord2action, action2coord, d4_transform, d4_inv_transform,
)

__version__ = "0.1"
from .selfplay import SelfPlay, MctsOptions, SpOptions  # noqa: F401,E402
from . import compat  # noqa: F401,E402
from .train import ReplayLoader, ReaderQueues, ReplayBuffer, parse_record, sgfstr_to_coords, coords_to_sgfstr  # noqa: F401,E402
from .client import ClientRecords, parse_request_seq, request_seq_to_json  # noqa: F401,E402


def tree_bytes_per_game(board_size, nodes_per_game):
    """HBM one game's search tree takes at `nodes_per_game` node ids (elfmcts_tree_bytes_per_game, include/elf_amd.h)."""
    return int(lib().elfmcts_tree_bytes_per_game(int(board_size), int(nodes_per_game)))


def mem_info(device=0):
    """(free, total) HBM bytes of a device (elfgo_mem_info)."""
    import ctypes as C
    fr, tot = C.c_size_t(0), C.c_size_t(0)
    from ._lib import check
    check(lib().elfgo_mem_info(int(device), C.byref(fr), C.byref(tot)))
    return int(fr.value), int(tot.value)
