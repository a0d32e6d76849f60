"""One-off randomized comparison of the GPU batch path with the oracle (longer and
wider than tests/test_gpu_parity.py::test_random_batches_equal_oracle).
usage: fuzz_gpu.py [seed] [seconds]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import oracle
hfz = importlib.import_module("huffman-avx512_amd")

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
D = hfz.distributions


def random_source():
    kind = int(rng.integers(0, 6))
    if kind == 0:
        return D.geometric_cdf()
    if kind == 1:
        return D.enwik_like_cdf()
    n = int(rng.integers(1, 257))
    w = [0] * 256
    idx = rng.choice(256, size=n, replace=False)
    if kind == 2:
        vals = rng.integers(1, 1000, size=n)
    elif kind == 3:
        vals = 1 << rng.integers(0, 40, size=n)
    elif kind == 4:
        vals = np.ones(n, dtype=np.int64)
    else:
        vals = (1.7 ** np.arange(n) % (1 << 40)).astype(np.int64) + 1
    for i, v in zip(idx, vals):
        w[int(i)] = int(v)
    return D.cdf_from_weights(w)


t0 = time.time()
case = 0
while time.time() - t0 < budget:
    K = int(rng.integers(1, 65))
    frame_bytes = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 20000)),
                                  int(rng.integers(20000, 300000))]))
    n_frames = int(rng.integers(1, 12))
    total = max(0, frame_bytes * n_frames - int(rng.integers(0, frame_bytes)))
    strided = bool(rng.integers(0, 2))
    bits, fpw, blk = int(rng.choice([0, 11, 12])), int(rng.choice([0, 1, 2])), int(rng.choice([0, 32, 64]))
    hfz.set_decoder_table_bits(bits)
    hfz.set_encoder_parts(int(rng.choice([0, 0, 1, 2, 8, 64])))
    hfz.set_decoder_store(int(rng.choice([0, 1, 2])))
    hfz.set_frames_per_wave(fpw)
    hfz.set_encoder_block(blk)
    raw = hfz.generate(random_source(), seed * 100003 + case, total) if total else torch.empty(
        0, dtype=torch.uint8, device="cuda")
    host = raw.cpu().numpy().tobytes()
    codec = hfz.BatchCodec(K, total, frame_bytes, strided=strided)
    batch = codec.compress(raw)
    out = codec.decompress(batch)
    torch.cuda.synchronize()
    tag = (case, K, frame_bytes, total, strided, bits, fpw, blk)
    assert int(codec.status.abs().sum().item()) == 0, tag
    assert int(codec.dstatus.abs().sum().item()) == 0, tag
    assert out.cpu().numpy().tobytes() == host, tag
    for i in range(codec.n_frames):
        assert batch.frame_bytes_host(i) == oracle.compress(
            K, host[i * frame_bytes:(i + 1) * frame_bytes]), tag + (i,)
    case += 1
    if case % 50 == 0:
        print("cases", case, "elapsed %.0f s" % (time.time() - t0), flush=True)
print("ok:", case, "cases, seed", seed)
