"""Shared fixtures for the CPU and GPU suites."""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
INPUTS = os.path.join(GOLDEN, "inputs")

# Stream counts the reference instantiates for the scalar path
# (codec/huffman.cpp:1977-1990) plus 64, the one-wave64 configuration.
KS = (1, 2, 4, 8, 16, 32, 64)


def load_input(name):
    with open(os.path.join(INPUTS, name), "rb") as f:
        return f.read()


def load_pins():
    with open(os.path.join(GOLDEN, "pins.json")) as f:
        return json.load(f)


def many_random():
    """The 100 inputs of codec/huffman_test.cpp:164-184 (ManyRandom)."""
    pack = load_input("many_random.pack")
    out, pos = [], 0
    while pos < len(pack):
        (n,) = struct.unpack_from("<I", pack, pos)
        pos += 4
        out.append(pack[pos:pos + n])
        pos += n
    assert len(out) == 100
    return out


# Every single-buffer case of the reference's codec test suite and benchmark
# generators (codec/huffman_test.cpp:56-184, codec/huffman_benchmark.cpp:61-216).
CASE_FILES = (
    "hello.bin", "lorem.bin", "equal_counts.bin", "bytes_x4.bin", "long_random.bin",
    "aaa.bin", "a1000.bin", "long_codes.bin", "empty.bin", "biased_100k.bin",
    "sorted_100k.bin", "uniform_100k.bin", "short_100.bin", "lorem_100k.bin",
)


def _dist():
    import importlib
    return importlib.import_module("huffman-avx512_amd").distributions


def geometric_cdf():
    return _dist().geometric_cdf()


def enwik_like_cdf():
    return _dist().enwik_like_cdf()
