"""Code-length statistics of the benchmark sources (CPU only): how often a code is longer than
a lookup table of B bits, how many symbols a lookup yields, and how often some lane of a
64-lane wave meets a long code in one lookup.  Codes come from the library's host code
builder; the sample is drawn with numpy from the source's probabilities (statistics only).
usage: code_stats.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hfz = importlib.import_module("huffman-avx512_amd")
rng = np.random.default_rng(7)
for name, frame in (("biased", 128 << 10), ("enwik", 512 << 10)):
    cdf = hfz.distributions.SOURCES[name]().astype(np.float64)
    p = np.diff(np.concatenate(([0.0], cdf))) / cdf[-1]
    buf = rng.choice(256, size=frame, p=p / p.sum()).astype(np.uint8)
    hist = np.bincount(buf, minlength=256).astype(np.uint32)
    lens = np.array(hfz.build_code_table(hist).len[:], dtype=int)
    q = hist / hist.sum()
    print("%s, one frame of %d KiB: %d symbols, mean code length %.3f bits" % (
        name, frame >> 10, int((hist > 0).sum()), float((q * lens).sum())))
    for lmin in (10, 11, 12):
        sel = (lens >= lmin) & (hist > 0)
        print("  codes of >= %d bits: %d symbols, %.3f %% of the bytes" % (lmin, int(sel.sum()), 100 * float(q[sel].sum())))
    sl = lens[buf[:200000]]
    for B in (9, 10, 11, 12):
        i = looks = esc = 0
        N = len(sl) - 2
        while i < N:
            looks += 1
            if sl[i] > B:
                esc += 1
                i += 1
            elif sl[i] + sl[i + 1] <= B:
                i += 2
            else:
                i += 1
        r = esc / looks
        print("  %2d-bit table: %.3f symbols per lookup, long code in %.2f %% of the lookups, in some lane of 64 in %.0f %%" % (
            B, N / looks, 100 * r, 100 * (1 - (1 - r) ** 64)))
