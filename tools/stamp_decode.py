"""Diagnostic: per-phase cycle shares of decode_kernel (needs tools/libhfz_stamp.so,
built with -DHFZ_STAMP).  Not part of the product or the tests.

Shares of phases that end in global stores (the encoder's tail, the code
builder's write-out) come out too large: every wave also flushes its timers with
atomics to a few addresses, and the stores queue behind them.  Check such a
phase with a build that skips it instead."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
hfz = importlib.import_module("huffman-avx512_amd")
hfz.LIB_PATH = os.path.join(ROOT, "tools", "libhfz_stamp.so")
L = hfz.lib()
geometric_cdf = hfz.distributions.geometric_cdf
names = ["header", "table", "priming", "staging", "rounds: loop", "rounds: lookups", "rounds: pair stores", "tail"]
SRC = os.environ.get("HFZ_SRC", "biased")
K = int(os.environ.get("HFZ_K", "64"))
for frame_kib in [int(a) for a in sys.argv[1:]] or [64]:
    total = int(os.environ.get("HFZ_MIB", "256")) << 20
    raw = hfz.generate(hfz.distributions.SOURCES[SRC](), 1000, total)
    codec = hfz.BatchCodec(K, total, frame_kib << 10)
    batch = codec.compress(raw)
    out = torch.empty(total, dtype=torch.uint8, device="cuda")
    codec.decompress(batch, out); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    L.hfz_debug_stamps(buf, 1)
    codec.compress(raw); codec.decompress(batch, out); torch.cuda.synchronize()
    L.hfz_debug_stamps(buf, 1)
    tot = sum(buf[:8]); nf = codec.n_frames
    print("frame_kib", frame_kib, "frames", nf, "cycles/wave %.0f" % (tot / nf),
          " ".join("%s=%.1f%%" % (n, 100.0 * buf[i] / tot) for i, n in enumerate(names)))
    et = sum(buf[8:16])
    print("  encode cycles/wave %.0f" % (et / nf), " ".join("%s=%.1f%%" % (n, 100.0 * buf[8 + i] / et)
          for i, n in enumerate(["table+header", "setup", "-", "main loop", "tail"])))
    bt = sum(buf[16:24])
    print("  build_code cycles/frame %.0f" % (bt / nf), " ".join("%s=%.1f%%" % (n, 100.0 * buf[16 + i] / max(bt, 1))
          for i, n in enumerate(["compact", "partition", "rank", "merge", "depths", "limit", "emit", "write-out"])))
    print("equal:", bool(torch.equal(out, raw)))
