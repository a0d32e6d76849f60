import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
hfz = importlib.import_module("huffman-avx512_amd")
import oracle
cdf = hfz.distributions.geometric_cdf()
for n, K, reps in ((100 << 10, 32, 200), (100 << 10, 4, 200), (64 << 20, 64, 5)):
    raw = oracle.generate(cdf, 3, n).tobytes()
    comp = hfz.compress(K, raw); back = hfz.decompress(K, comp); assert back == raw
    t0 = time.perf_counter()
    for _ in range(reps): comp = hfz.compress(K, raw)
    t1 = time.perf_counter()
    for _ in range(reps): back = hfz.decompress(K, comp)
    t2 = time.perf_counter()
    m = reps * n / float(1 << 20)
    print("host-buffer API n=%d K=%d: compress %.1f MiB/s, decompress %.1f MiB/s" % (n, K, m / (t1 - t0), m / (t2 - t1)))
