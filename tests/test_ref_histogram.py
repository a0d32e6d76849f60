"""The oracle's histogram against the reference's own MakeHistogram, compiled
from the unmodified codec/histogram.cpp (oracle/_ref/libref_hist.so, recipe in
oracle/Makefile target `ref`).  This pins ho_histogram (and through it the
device histogram kernels, which the GPU tests compare with ho_histogram) to the
reference itself rather than to recorded values.

Runs only where the compiled reference exists (it is built in the container that
holds /root/reference) and the CPU has the AVX-512 subsets histogram.cpp uses.
"""
import os

import numpy as np
import pytest

import oracle
from helpers import CASE_FILES, INPUTS, load_input, many_random

pytestmark = pytest.mark.skipif(not oracle.ref_available(),
                                reason="oracle/_ref/libref_hist.so not built or CPU lacks AVX-512")

VARIANTS = sorted(oracle.REF_HIST_VARIANTS)


def _all_fixture_files():
    return sorted(f for f in os.listdir(INPUTS) if f.endswith(".bin"))


@pytest.mark.parametrize("name", _all_fixture_files())
def test_every_fixture_every_variant(name):
    raw = load_input(name)
    want = oracle.ref_histogram(raw)  # the dispatcher, codec/histogram.cpp:193-201
    assert int(want.sum()) == len(raw)
    assert np.array_equal(oracle.histogram(raw), want), name
    for v in VARIANTS:
        assert np.array_equal(oracle.ref_histogram(raw, v), want), (name, v)


def test_reference_histogram_test_cases():
    # codec/histogram_test.cpp:18-26 (ShortSanity) and :28-42 (Long, 2007 bytes)
    h = oracle.ref_histogram(b"foobar")
    assert [int(h[ord(c)]) for c in "fobar"] == [1, 2, 1, 1, 1] and int(h.sum()) == 6
    assert np.array_equal(oracle.histogram(b"foobar"), h)
    raw = load_input("hist_long.bin")
    assert len(raw) == 2007
    for v in VARIANTS:
        assert np.array_equal(oracle.ref_histogram(raw, v), oracle.histogram(raw)), v


def test_many_random_and_slices():
    # the 100 short inputs of ManyRandom, and the slice shapes CompressMulti<K> feeds
    # MakeHistogram with (codec/huffman.cpp:755-761): around the 1500-byte dispatch
    # threshold, unaligned starts, odd lengths
    for raw in many_random():
        assert np.array_equal(oracle.histogram(raw), oracle.ref_histogram(raw))
    big = np.frombuffer(load_input("long_random.bin"), dtype=np.uint8)
    for lo, n in ((0, 1499), (1, 1500), (3, 1501), (5, 3125), (7, 3200), (9, 25600), (11, 65535),
                  (13, 99987), (0, 0), (17, 1), (2, 15), (4, 16), (6, 17)):
        part = big[lo:lo + n]
        want = oracle.ref_histogram(part)
        assert np.array_equal(oracle.histogram(part), want), (lo, n)
        for v in VARIANTS:
            assert np.array_equal(oracle.ref_histogram(part, v), want), (lo, n, v)


def test_slice_histograms_sum_to_frame_histogram():
    # CompressMulti<K> sums K per-slice histograms (codec/huffman.cpp:755-767)
    raw = load_input("biased_100k.bin")
    for K in (4, 32, 64):
        sizes = oracle.slice_sizes(len(raw), K)
        total = np.zeros(256, dtype=np.uint64)
        pos = 0
        for s in sizes:
            total += oracle.ref_histogram(raw[pos:pos + int(s)]).astype(np.uint64)
            pos += int(s)
        assert pos == len(raw)
        assert np.array_equal(total.astype(np.uint32), oracle.histogram(raw)), K
