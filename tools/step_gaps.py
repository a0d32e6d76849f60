"""How much of a step the HIP events between kernels cost (config 2 batch)."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
import torch
hfz = importlib.import_module("huffman-avx512_amd")
total, fb, K = 256 << 20, 256 << 10, 64
raw = hfz.generate(hfz.distributions.geometric_cdf(), 1000, total)
codec = hfz.BatchCodec(K, total, fb)
out = torch.empty(total, dtype=torch.uint8, device="cuda")
def run(timing, steps=40):
    for _ in range(3):
        b = codec.compress(raw); codec.decompress(b, out)
    torch.cuda.synchronize()
    if timing: hfz.timing_enable(timing)
    t0 = time.perf_counter()
    for _ in range(steps):
        b = codec.compress(raw); codec.decompress(b, out)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if timing:
        kt = hfz.timing_read(); hfz.timing_enable(False)
        print("kernel sum %.4f" % sum(kt[k + "_ms"] for k in ("hist", "table", "sizes", "place", "encode", "decode")))
    return 1e3 * (t1 - t0) / steps
for _ in range(2):
    print("timing off: %.4f ms/step" % run(False))
    print("timing on : %.4f ms/step" % run(1))
    print("every 4th : %.4f ms/step" % run(4))
