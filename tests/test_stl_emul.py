"""CPU: elf_amd/csrc/stl_emul.h (the code the expand kernel runs on the GPU) against the real libstdc++:
iteration order of std::unordered_map<unsigned short, ...> after 0..441 insertions, and std::sort with the
reference's comparator (go/mcts/mcts.h:292-297) on inputs with heavy ties and adversarial patterns."""
import os
import subprocess

from conftest import ROOT


def test_stl_emulation_matches_libstdcxx(tmp_path):
    exe = str(tmp_path / "stl_emul_check")
    src = os.path.join(ROOT, "tests", "native", "stl_emul_check.cc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout
    assert "bad 0" in r.stdout.splitlines()[0] and "bad 0" in r.stdout.splitlines()[1], r.stdout
