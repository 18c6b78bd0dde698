"""The reference's module names resolve to this library's pybind11 extensions (elf_amd/compat.py); SGF main-line reader; GTP
coordinate letters.  The boundary itself is exercised in tests/test_pybind_boundary.py."""
import glob

import pytest


def test_reference_module_names_resolve(built):
    """the import lines of src_py/elfgames/go/game_inference.py:15, game.py:15 and src_py/elf/__init__.py:8"""
    from elf_amd import compat
    mods = compat.install_reference_module_names()
    import _elfgames_go as go2
    import _elfgames_go_inference as go
    from _elf import SearchAlgoOptions, TSOptions   # noqa: F401
    assert [m.__name__ for m in mods] == ["_elf", "_elfgames_go_inference", "_elfgames_go"]
    assert all(m.__file__.endswith(".so") for m in mods)          # real extensions, not Python shims
    co, opt = go.ContextOptions(), go.GameOptions()
    co.mcts_options.alg_opt.c_puct = 1.5
    assert isinstance(co.mcts_options, TSOptions) and go2.ContextOptions is go.ContextOptions and go2.GameContext is not go.GameContext
    co.batchsize, opt.mode = 8, "online"
    co.mcts_options.num_threads = 1
    GC = go.GameContext(co, opt)
    assert GC.getParams()["num_action"] == 362 and GC.getParams()["ACTION_PASS"] == -99


def test_sgf_main_line_matches_reference_loader(built):
    """the preload_sgf reader of the boundary against the reference's own Sgf loader (via oracle/_ref) on ladder-suite files"""
    from elf_amd import compat
    from pyoracle import Ref
    compat.install_reference_module_names()
    import _elf
    files = sorted(glob.glob("/root/reference/ladder_suite/ladder/*.sgf"))[:25]
    if not files or not Ref.available(19):
        pytest.skip("reference tree / oracle/_ref not present")
    R = Ref(19)
    for f in files:
        mv, _ = R.sgf_moves(f)
        assert list(_elf._go.sgf_main_line(f, 19)) == [int(c) for c in mv], f


def test_gtp_coordinate_letters(built):
    """console_lib.py:12-29: GTP letters skip 'I'; round trip over the whole 19x19 board"""
    from elf_amd.gtp import move2xy, xy2move
    assert xy2move(0, 0) == "A1" and xy2move(7, 3) == "H4" and xy2move(8, 3) == "J4" and xy2move(18, 18) == "T19"
    assert move2xy("pass") == (-1, -1) and xy2move(-1, -1) == "pass"
    for x in range(19):
        for y in range(19):
            m = xy2move(x, y)
            assert "I" not in m and move2xy(m) == (x, y) and move2xy(m.lower()) == (x, y)
