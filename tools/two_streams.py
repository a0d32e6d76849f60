"""Diagnostic: throughput of independent batches issued on one stream versus on
two streams (what a caller with a queue of batches can do: the serial code
construction of one batch then overlaps the memory-bound passes of the other).
usage: two_streams.py [total_mib] [frame_kib] [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
hfz = importlib.import_module("huffman-avx512_amd")

total = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
fb = (int(sys.argv[2]) if len(sys.argv) > 2 else 128) << 10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
K = 64
cdf = hfz.distributions.geometric_cdf()
raws = [hfz.generate(cdf, 1000 + i, total) for i in range(2)]
codecs = [hfz.BatchCodec(K, total, fb) for _ in range(2)]
outs = [torch.empty(total, dtype=torch.uint8, device="cuda") for _ in range(2)]


def run(streams):
    for i in range(2):  # warm-up
        with torch.cuda.stream(streams[i % len(streams)]):
            codecs[i].decompress(codecs[i].compress(raws[i]), outs[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        i = s & 1
        with torch.cuda.stream(streams[i % len(streams)]):
            codecs[i].decompress(codecs[i].compress(raws[i]), outs[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.equal(outs[0], raws[0]) and torch.equal(outs[1], raws[1])
    return steps * (total >> 20) / dt, dt / steps * 1e3


one = [torch.cuda.Stream()]
two = [torch.cuda.Stream(), torch.cuda.Stream()]
for name, st in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    mibps, ms = run(st)
    print("%-12s %8.0f MiB/s  %.4f ms per batch" % (name, mibps, ms), flush=True)
