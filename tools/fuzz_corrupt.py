"""One-off: random damage to compressed batches (bytes, frame offsets, frame sizes);
the batch decoder has to report or decode garbage, never fault or hang.
usage: fuzz_corrupt.py [seed] [seconds]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
hfz = importlib.import_module("huffman-avx512_amd")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
D = hfz.distributions
sources = [D.geometric_cdf(), D.enwik_like_cdf(), D.cdf_from_weights([1] * 256),
           D.cdf_from_weights([1 << (40 - i) for i in range(40)] + [0] * 216)]
t0 = time.time()
case = rejected = 0
while time.time() - t0 < budget:
    K = int(rng.choice([1, 3, 8, 16, 31, 32, 33, 64]))
    frame_bytes = int(rng.choice([64, 1000, 9000, 70000]))
    n_frames = int(rng.integers(1, 7))
    total = frame_bytes * n_frames - int(rng.integers(0, frame_bytes))
    hfz.set_decoder_table_bits(int(rng.choice([0, 11, 12])))
    hfz.set_frames_per_wave(int(rng.choice([0, 1, 2])))
    hfz.set_decoder_store(int(rng.choice([0, 1, 2])))
    raw = hfz.generate(sources[int(rng.integers(0, len(sources)))], seed * 7919 + case, total)
    codec = hfz.BatchCodec(K, total, frame_bytes, strided=bool(rng.integers(0, 2)))
    batch = codec.compress(raw)
    torch.cuda.synchronize()
    used = int((batch.comp_off[-1] + batch.comp_size[-1].to(torch.int64)).item())
    comp = batch.comp.cpu().numpy()
    offs = batch.comp_off.cpu().numpy().copy()
    sizes = batch.comp_size.cpu().numpy().copy().view(np.uint32)
    mode = int(rng.integers(0, 5))
    if mode <= 2:  # byte damage, biased towards the frame headers
        for _ in range(int(rng.integers(1, 20))):
            f = int(rng.integers(0, len(offs)))
            span = int(sizes[f]) if mode == 0 else min(int(sizes[f]), 300)
            pos = int(offs[f]) + int(rng.integers(0, max(span, 1)))
            if pos < comp.size:
                comp[pos] = int(rng.integers(0, 256))
    elif mode == 3:  # frame size table
        f = int(rng.integers(0, len(sizes)))
        sizes[f] = int(rng.choice([0, 7, 8, 9, int(sizes[f]) // 2, int(sizes[f]) + 1000, 0xffffffff]))
    else:  # frame offset table
        f = int(rng.integers(0, len(offs)))
        offs[f] = int(rng.choice([0, 1, int(offs[f]) + 3, used, comp.size - 1, comp.size, 1 << 40, (1 << 63) - 1]))
    batch.comp.copy_(torch.from_numpy(comp))
    batch.comp_off.copy_(torch.from_numpy(offs))
    batch.comp_size.copy_(torch.from_numpy(sizes.view(np.int32)))
    codec.decompress(batch, shared=bool(rng.integers(0, 4) == 0))
    torch.cuda.synchronize()
    rejected += int((codec.dstatus != 0).any().item())
    case += 1
    if case % 200 == 0:
        print("cases", case, "with a rejected frame", rejected, "elapsed %.0f s" % (time.time() - t0), flush=True)
hfz.set_decoder_table_bits(0)
hfz.set_frames_per_wave(0)
# the device is still usable
raw = hfz.generate(sources[0], 1, 1 << 20)
codec = hfz.BatchCodec(64, 1 << 20, 65536)
assert torch.equal(codec.decompress(codec.compress(raw)), raw)
print("ok:", case, "cases,", rejected, "with a rejected frame, seed", seed)
