"""Where the reference's pybind11 module names come from in this repository.

The reference's Python imports `_elf` (src_py/elf/__init__.py:8), `_elfgames_go` (src_py/elfgames/go/game.py:15) and
`_elfgames_go_inference` (game_inference.py:15).  Here they are real pybind11 extensions (elf_amd/csrc/pybind_elf.cc,
pybind_go_modules.cc, built into elf_amd/ext/ by `make -C elf_amd/csrc`) with the names of src_cpp/elf/Pybind.cc:27-117,
elfgames/go/train/Pybind.cc:18-63 and elfgames/go/inference/Pybind.cc:18-45, over the C ABI of libelf_amd.so.  This module only
puts that directory on sys.path; there are no Python look-alikes of the C++ classes.

    import elf_amd.compat as compat; compat.install_reference_module_names()
    import _elfgames_go as go            # go.ContextOptions(), go.GameOptions(), go.GameContext(co, opt)
    from elf.utils_elf import GCWrapper  # the reference's own, unmodified
"""
import importlib
import os
import sys

EXT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ext")
MODULES = ("_elf", "_elfgames_go_inference", "_elfgames_go")


def install_reference_module_names():
    """Make `import _elf`, `import _elfgames_go`, `import _elfgames_go_inference` resolve to this library's extensions.
    Raises ImportError (no fallback) when they have not been built."""
    if EXT_DIR not in sys.path:
        sys.path.insert(0, EXT_DIR)
    return [importlib.import_module(m) for m in MODULES]
