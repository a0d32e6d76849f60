"""Diagnostic: cycle stamps of the decoder on ONE frame (the several-threads-per-stream build),
thread 0's wave (needs tools/libhfz_stamp.so, built with -DHFZ_STAMP).  Not part of the product."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
hfz = importlib.import_module("huffman-avx512_amd")
hfz.LIB_PATH = os.path.join(ROOT, "tools", "libhfz_stamp.so")
L = hfz.lib()
names = ["header", "table", "priming(+prepass)", "staging", "rounds: loop", "rounds: lookups", "rounds: pair stores", "tail"]
for n, K in ((100 << 10, 4), (100 << 10, 64), (1 << 20, 64)):
    raw = hfz.generate(hfz.distributions.geometric_cdf(), 3, n)
    codec = hfz.BatchCodec(K, n, n)
    batch = codec.compress(raw)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    codec.decompress(batch, out); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    L.hfz_debug_stamps(buf, 1)
    codec.decompress(batch, out); torch.cuda.synchronize()
    L.hfz_debug_stamps(buf, 1)
    tot = sum(buf[:8])
    print("n", n, "K", K, "cycles (thread 0's wave) %d" % tot, " ".join("%s=%d" % (nm, buf[i]) for i, nm in enumerate(names)), "equal", bool(torch.equal(out, raw)))
