"""The CPU oracle against the known-answer pins of the reference.

The reference holds no byte-level golden vectors of its own (its tests are
round-trip only, codec/huffman_test.cpp:56-184), so the pins are outputs of the
compiled reference recorded once (tests/golden/pins.json).
"""
import numpy as np
import pytest

import oracle
from helpers import CASE_FILES, KS, load_input, load_pins, many_random

PINS = load_pins()


@pytest.mark.parametrize("name", sorted(PINS["inputs"]))
def test_compress_matches_reference_pins(name):
    entry = PINS["inputs"][name]
    raw = load_input(name)
    assert len(raw) == entry["raw_len"]
    if "raw_hash" in entry:
        assert oracle.pin_hash(raw) == entry["raw_hash"]
    for K, (size, h) in entry["K"].items():
        comp = oracle.compress(int(K), raw)
        assert len(comp) == size, (name, K)
        assert oracle.pin_hash(comp) == h, (name, K)


def test_byte_pins():
    bp = PINS["byte_pins"]
    assert oracle.compress(2, b"Hello World").hex() == bp["hello_K2"]
    assert oracle.compress(1, b"AAA").hex() == bp["aaa_K1"]
    lc = oracle.compress(1, load_input("long_codes.bin")).hex()
    assert lc.startswith(bp["long_codes_header_prefix"].replace(" ", ""))
    bi = oracle.compress(4, load_input("biased_100k.bin")).hex()
    assert bi.startswith(bp["biased_100k_header_prefix"].replace(" ", ""))


@pytest.mark.parametrize("name", CASE_FILES)
def test_round_trip_reference_cases(name):
    # codec/huffman_test.cpp: every assertion is Decompress(Compress(x)) == x
    raw = load_input(name)
    for K in KS:
        comp = oracle.compress(K, raw)
        assert oracle.decompress(K, comp) == raw, (name, K)


def test_round_trip_many_random():
    for i, raw in enumerate(many_random()):
        for K in (1, 4, 8, 32, 64):
            assert oracle.decompress(K, oracle.compress(K, raw)) == raw, (i, K)


def test_histogram_cases():
    # codec/histogram_test.cpp:18-42
    h = oracle.histogram(b"foobar")
    assert [h[ord(c)] for c in "fobarq"] == [1, 2, 1, 1, 1, 0]
    raw = load_input("hist_long.bin")
    assert np.array_equal(oracle.histogram(raw),
                          np.bincount(np.frombuffer(raw, dtype=np.uint8), minlength=256))


def test_frame_layout_properties():
    # first 8 bytes of every stream region are zero; sizes follow the header
    raw = load_input("biased_100k.bin")
    for K in (4, 32, 64):
        comp = oracle.compress(K, raw)
        mask = int.from_bytes(comp[4:8], "little")
        nlen = bin(mask).count("1")
        counts = comp[8:8 + nlen]
        nsyms = sum(counts) if not (nlen == 1 and counts[0] == 0) else 256
        hdr = 8 + nlen + nsyms + 4 * (K - 1)
        ends = [int.from_bytes(comp[8 + nlen + nsyms + 4 * k:][:4], "little") for k in range(K - 1)]
        ends.append(len(comp) - hdr)
        start = 0
        for e in ends:
            assert comp[hdr + start:hdr + start + 8] == bytes(8)
            assert e - start >= 8
            start = e
