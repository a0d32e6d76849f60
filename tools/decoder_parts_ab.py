"""Decode time by threads per stream (hfz_set_decoder_parts) on batches of few
frames, and the host-buffer entry points on single frames.  Prints a table."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
hfz = importlib.import_module("huffman-avx512_amd")
import oracle

def decode_ms(codec, batch, out, reps=10):
    for _ in range(3):
        codec.decompress(batch, out)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        codec.decompress(batch, out)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps

pr = torch.cuda.get_device_properties(0)
print("LDS per block:", getattr(pr, "shared_memory_per_block", None), "optin:", getattr(pr, "shared_memory_per_block_optin", None), flush=True)
cases = [("biased", 64, 256, 1024), ("biased", 64, 128, 1024), ("biased", 64, 256, 256), ("biased", 64, 256, 512),
         ("biased", 64, 64, 1024), ("enwik", 32, 256, 1024), ("biased", 64, 8, 1024), ("biased", 4, 1, 100), ("biased", 64, 256, 128)]
for source, K, total_mib, frame_kib in cases:
    total, frame = total_mib << 20, frame_kib << 10
    raw = hfz.generate(hfz.distributions.SOURCES[source](), 5, total)
    codec = hfz.BatchCodec(K, total, frame)
    batch = codec.compress(raw)
    out = torch.empty(total, dtype=torch.uint8, device="cuda")
    row = []
    for parts in (1, 2, 4, 8, 16, 32, 64, 0):
        hfz.set_decoder_parts(parts)
        ms = decode_ms(codec, batch, out)
        assert torch.equal(out, raw), (source, K, parts)
        assert int(codec.dstatus.abs().sum().item()) == 0
        row.append("%s=%.4f" % ("auto" if parts == 0 else "p%d" % parts, ms))
    hfz.set_decoder_parts(0)
    print("%s K=%d %d MiB in %d KiB frames (%d frames): decode ms %s" % (
        source, K, total_mib, frame_kib, codec.n_frames, " ".join(row)), flush=True)

cdf = hfz.distributions.geometric_cdf()
for n, K, reps in ((100 << 10, 32, 300), (100 << 10, 4, 300), (1 << 20, 64, 100), (64 << 20, 64, 5)):
    raw = oracle.generate(cdf, 3, n).tobytes()
    comp = hfz.compress(K, raw); back = hfz.decompress(K, comp); assert back == raw
    t0 = time.perf_counter()
    for _ in range(reps): comp = hfz.compress(K, raw)
    t1 = time.perf_counter()
    for _ in range(reps): back = hfz.decompress(K, comp)
    t2 = time.perf_counter()
    m = reps * n / float(1 << 20)
    print("host-buffer API n=%d K=%d: compress %.1f MiB/s (%.1f us), decompress %.1f MiB/s (%.1f us)" % (
        n, K, m / (t1 - t0), 1e6 * (t1 - t0) / reps, m / (t2 - t1), 1e6 * (t2 - t1) / reps), flush=True)
