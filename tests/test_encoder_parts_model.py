"""CPU model of the several-threads-per-stream encoder: the byte ranges and lead
bits every thread of encode_kernel<.., kParts = true> works with, replayed in
Python against the oracle's frames.  Pins the arithmetic of the split (first bit
of a part, ownership of the bytes where parts meet, lead bits fetched from the
symbols before the part) without a GPU."""
import numpy as np
import pytest

import oracle
from helpers import load_input, many_random


def slice_of(n, K, k):
    q, r = divmod(n, K)
    return k * q + min(k, r), q + (1 if k < r else 0)


def model_frame(K, P, raw):
    raw = bytes(raw)
    frame = oracle.compress(K, raw)
    n = len(raw)
    coding = oracle.make_coding(oracle.histogram(raw))
    clen = [int(x) for x in coding.len]
    cbits = [int(x) for x in coding.bits]  # left-aligned in 12 bits
    nlen = bin(coding.len_mask).count("1")
    hdr_total = 8 + nlen + coding.num_syms + 4 * (K - 1)
    eo = 8 + nlen + coding.num_syms
    ends = [int.from_bytes(frame[eo + 4 * k:eo + 4 * k + 4], "little") for k in range(K - 1)]
    ends.append(len(frame) - hdr_total)
    out = bytearray(len(frame))
    out[:hdr_total] = frame[:hdr_total]
    owner = [None] * len(frame)
    for k in range(K):
        s_begin, s_size = slice_of(n, K, k)
        stream_hi = hdr_total + ends[k]
        stream_lo = hdr_total + (ends[k - 1] if k else 0)
        part_bits = []
        for p in range(P):
            b, sz = slice_of(s_size, P, p)
            part_bits.append(sum(clen[c] for c in raw[s_begin + b:s_begin + b + sz]))
        incl = 0
        for p in range(P):
            first_bit = incl
            incl += part_bits[p]
            r_hi = stream_hi - (first_bit >> 3)
            r_lo = stream_lo if p == P - 1 else stream_hi - (incl >> 3)
            assert stream_lo <= r_lo <= r_hi <= stream_hi
            b, sz = slice_of(s_size, P, p)
            in_lo = s_begin + b
            lead_n = first_bit & 7
            lead, have, q = 0, 0, in_lo
            while have < lead_n and q > s_begin:
                q -= 1
                L = clen[raw[q]]
                if L:
                    lead |= (cbits[raw[q]] >> (12 - L)) << have
                have += L
            assert have >= lead_n
            lead &= (1 << lead_n) - 1
            if r_lo >= r_hi:
                continue
            # the writer: lead bits, the part's codes, padding to a byte, 8 zero bytes
            bits = [(lead >> (lead_n - 1 - i)) & 1 for i in range(lead_n)]
            for c in raw[in_lo:in_lo + sz]:
                L = clen[c]
                v = cbits[c] >> (12 - L) if L else 0
                bits += [(v >> (L - 1 - i)) & 1 for i in range(L)]
            bits += [0] * ((-len(bits)) % 8) + [0] * 64
            addr = r_hi - 1
            for i in range(0, len(bits), 8):
                if r_lo <= addr < r_hi:
                    v = 0
                    for bit in bits[i:i + 8]:
                        v = (v << 1) | bit
                    assert owner[addr] is None, (k, p, addr, owner[addr])
                    owner[addr] = (k, p)
                    out[addr] = v
                addr -= 1
    assert all(o is not None for o in owner[hdr_total:]), "a stream byte nobody wrote"
    return bytes(out), frame


@pytest.mark.parametrize("P", [2, 4, 64])
def test_parts_tile_the_streams_and_reproduce_the_frame(P):
    cases = [load_input("hello.bin"), load_input("lorem.bin"), load_input("a1000.bin"), load_input("equal_counts.bin"),
             load_input("long_codes.bin")[:20000], load_input("biased_100k.bin")[:30000], b"", b"AAA"]
    cases += many_random()[:12]
    for raw in cases:
        for K in (1, 3, 4, 32, 64):
            got, want = model_frame(K, P, raw)
            assert got == want, (P, K, len(raw))
