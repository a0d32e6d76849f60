"""Where the time of a single-frame call goes: kernel times of a one-frame batch (device
buffers, hfz_timing) next to the whole host-buffer call.  usage: single_frame_breakdown.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
hfz = importlib.import_module("huffman-avx512_amd")
cdf = hfz.distributions.geometric_cdf()
for n, K in ((100 << 10, 4), (100 << 10, 32), (100 << 10, 64), (1 << 20, 64)):
    raw = hfz.generate(cdf, 3, n)
    codec = hfz.BatchCodec(K, n, n)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        b = codec.compress(raw); codec.decompress(b, out)
    torch.cuda.synchronize()
    hfz.timing_enable(True)
    for _ in range(50):
        b = codec.compress(raw); codec.decompress(b, out)
    torch.cuda.synchronize()
    kt = hfz.timing_read(); hfz.timing_enable(False)
    host = raw.cpu().numpy().tobytes()
    comp = hfz.compress(K, host); assert hfz.decompress(K, comp) == host
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps): comp = hfz.compress(K, host)
    t1 = time.perf_counter()
    for _ in range(reps): back = hfz.decompress(K, comp)
    t2 = time.perf_counter()
    print("n=%d K=%d: kernels us count %.1f code %.1f place %.1f encode %.1f | decode %.1f ; host calls us compress %.1f decompress %.1f" % (
        n, K, 1e3 * kt["hist_ms"], 1e3 * kt["table_ms"], 1e3 * kt["place_ms"], 1e3 * kt["encode_ms"], 1e3 * kt["decode_ms"],
        1e6 * (t1 - t0) / reps, 1e6 * (t2 - t1) / reps), flush=True)
