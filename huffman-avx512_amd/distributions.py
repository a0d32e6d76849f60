"""Integer-only 256-symbol sources for the benchmark and the tests.

A source is a table cdf[256] of uint32 thresholds, cdf[j] = floor(2^32 * P(X <= j));
byte i of a stream is the number of j < 255 with cdf[j] <= u_i, where u_i is the
high half of a counter-based splitmix64 draw (hfz_generate_dev in include/hfz.h,
ho_generate in the oracle).  No floating point is involved in sampling, so host
and device produce identical bytes.
"""
import numpy as np


def cdf_from_weights(weights):
    w = [int(x) for x in weights]
    assert len(w) == 256 and min(w) >= 0 and sum(w) > 0
    total, acc, out = sum(w), 0, []
    for x in w:
        acc += x
        out.append(min((acc << 32) // total, 0xFFFFFFFF))
    out[255] = 0xFFFFFFFF
    return np.array(out, dtype=np.uint32)


def geometric_cdf(p_num=1, p_den=5):
    """P(i) = (1-p)^i * p: the biased source of the reference's benchmark
    (GenerateProbaData(0.2), codec/huffman_benchmark.cpp:27-36), as exact
    integer thresholds floor(2^32 * (1 - (1-p)^(j+1)))."""
    q_num, q_den = p_den - p_num, p_den
    out = []
    for j in range(256):
        a, b = q_num ** (j + 1), q_den ** (j + 1)
        out.append(min(((b - a) << 32) // b, 0xFFFFFFFF))
    out[255] = 0xFFFFFFFF
    return np.array(out, dtype=np.uint32)


def _enwik_weights():
    """Order-0 byte weights (per ~100000) imitating English-Wikipedia XML text.

    Provenance: hand-built for this repository from general knowledge of English
    letter frequencies and wiki/XML markup (brackets, quotes, entities, digits,
    UTF-8 lead/continuation bytes).  It is NOT measured from enwik8, which is not
    available offline; the reference reads that file from argv[1]
    (codec/huffman_benchmark.cpp:38-59).  Order-0 entropy of this table is about
    5.1 bits/byte, close to enwik8's order-0 entropy.  All 256 byte values get a
    nonzero weight so that the 12-bit length limit is exercised.
    """
    w = [1] * 256
    def put(chars, value):
        for ch in chars:
            w[ord(ch) if isinstance(ch, str) else ch] = value
    put(" ", 17000)
    for ch, v in zip("etainorsl", (9400, 6900, 6600, 5800, 5700, 5500, 5000, 4800, 3200)):
        put(ch, v)
    for ch, v in zip("hdcumpgfybwvkxjqz", (2800, 2600, 2500, 1900, 1800, 1600, 1500, 1400, 1100,
                                           1050, 950, 700, 450, 150, 100, 150, 80)):
        put(ch, v)
    put("\n", 1100); put(",.", 900); put("[]", 1250); put("'", 800); put("|", 600)
    put("=", 550); put(";", 500); put("&", 450); put("/", 450); put("<>", 700); put(":", 400)
    put("-", 350); put('"', 300); put("()", 200); put("*", 150); put("{}", 80); put("#", 60)
    put("_", 100); put("%", 40); put("!", 30); put("?+", 20); put("\t", 20)
    put("1", 800); put("0", 750); put("9", 500); put("2", 450); put("345678", 250)
    for ch, v in zip("ASCTIMBPRDELFHGNWUOJKVYQXZ", (350, 330, 320, 300, 270, 260, 220, 230, 200, 190,
                                                     200, 180, 170, 170, 160, 170, 150, 110, 120, 100,
                                                     90, 70, 40, 20, 20, 20)):
        put(ch, v)
    for b in range(0x80, 0xC0):
        w[b] = 10
    for b in range(0xC2, 0xE0):
        w[b] = 8
    for b in range(0xE0, 0xF0):
        w[b] = 6
    w[0xC3], w[0xE2], w[0x80] = 120, 80, 80
    return w


def enwik_like_cdf():
    return cdf_from_weights(_enwik_weights())


def entropy_bits(weights):
    p = np.array(weights, dtype=np.float64)
    p = p / p.sum()
    p = p[p > 0]
    return float(-(p * np.log2(p)).sum())


SOURCES = {"biased": geometric_cdf, "enwik": enwik_like_cdf}
