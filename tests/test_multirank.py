"""N > 1 path on CPU: two gloo ranks shard one batch by frame ranges.

Frames are independent, so the only distributed logic is the partition itself
and the bookkeeping bench.py does (barrier, max-over-ranks time).  Each rank
compresses its frames (with the CPU oracle standing in for the GPU here, there
is no GPU in this container) and rank 0 checks that the concatenation of the
shards equals the single-rank result frame for frame.
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle
    from helpers import geometric_cdf
    hfz = importlib.import_module("huffman-avx512_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, frame_bytes, total = 32, 4096, 4096 * 11 + 123  # 12 frames, ragged tail
    raw = oracle.generate(geometric_cdf(), 5, total).tobytes()
    n_frames = hfz.lib().hfz_batch_frames(total, frame_bytes)
    lo, hi = hfz.shard_frames(n_frames, rank, world)
    mine = [oracle.compress(K, raw[i * frame_bytes:(i + 1) * frame_bytes]) for i in range(lo, hi)]
    # what bench.py does: barrier, then the slowest rank's time
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, mine))
    if rank == 0:
        frames, cover = [], []
        for lo_r, hi_r, fr in sorted(gathered):
            cover += list(range(lo_r, hi_r))
            frames += fr
        assert cover == list(range(n_frames))
        want = [oracle.compress(K, raw[i * frame_bytes:(i + 1) * frame_bytes])
                for i in range(n_frames)]
        assert frames == want
        back = b"".join(oracle.decompress(K, f) for f in frames)
        assert back == raw
        open(os.path.join(tmpdir, "ok"), "w").write("ok")
    dist.destroy_process_group()


def _shared_code_worker(rank, world, port, tmpdir):
    # config "one code for the whole job": every rank counts its own shard, the
    # 256 counters are summed over ranks (the job's only exchange), every rank
    # builds the code from the sum and codes its frames with it
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle
    hfz = importlib.import_module("huffman-avx512_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, frame_bytes, total = 16, 5000, 5000 * 6 + 77
    cdf = hfz.distributions.geometric_cdf() if rank == 0 else hfz.distributions.enwik_like_cdf()
    raw = oracle.generate(cdf, 100 + rank, total).tobytes()  # every rank its own data
    h = torch.from_numpy(oracle.histogram(raw).astype(np.int64))
    dist.all_reduce(h)
    h32 = hfz.scale_histogram(h.numpy())
    record = bytes(hfz.frame_code(h32))            # host builder of the library
    coding = oracle.make_coding(h32)               # the oracle's builder
    frames = [oracle.compress_with_coding(K, coding, raw[i:i + frame_bytes])
              for i in range(0, total, frame_bytes)]
    gathered = [None] * world
    dist.all_gather_object(gathered, (record, bytes(h32.tobytes()), frames, raw))
    if rank == 0:
        assert all(g[0] == gathered[0][0] and g[1] == gathered[0][1] for g in gathered)
        fc = hfz.frame_code(h32)
        code_section = bytes(fc.hdr)[:fc.hdr_len]
        for _, _, frs, data in gathered:
            for i, fr in enumerate(frs):
                assert fr[4:4 + len(code_section)] == code_section  # same code in every frame
                assert oracle.decompress(K, fr) == data[i * frame_bytes:(i + 1) * frame_bytes]
        # counts beyond 32 bits keep every occurring symbol
        big = np.zeros(256, dtype=np.uint64)
        big[0], big[1], big[2] = 1 << 40, 3, 0
        s = hfz.scale_histogram(big)
        assert s[0] == 1 << 31 and s[1] == 1 and s[2] == 0
        open(os.path.join(tmpdir, "ok2"), "w").write("ok")
    dist.destroy_process_group()


def test_two_ranks_shared_code(tmp_path):
    import torch.multiprocessing as mp
    port = 30100 + (os.getpid() % 500)
    mp.spawn(_shared_code_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok2").exists()


def test_two_ranks_shard_frames(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_frames_partition():
    hfz = importlib.import_module("huffman-avx512_amd")
    for n in (0, 1, 7, 8, 9, 4096, 65536):
        for world in (1, 2, 3, 8):
            ranges = [hfz.shard_frames(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        hfz.shard_frames(4, 2, 2)
