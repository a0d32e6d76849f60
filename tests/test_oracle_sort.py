"""The oracle's restated libstdc++ introsort against std::sort itself."""
import os
import subprocess

from helpers import HERE


def test_sort_matches_std_sort(tmp_path):
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "sort_check")
    obj = str(tmp_path / "ho.o")
    subprocess.check_call(["gcc", "-O2", "-c", os.path.join(root, "oracle", "huffman_oracle.c"),
                           "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(HERE, "native", "sort_check.cpp"),
                           obj, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.strip().endswith("OK")
