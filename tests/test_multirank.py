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
        assert s[0] == 1 << 30 and s[1] == 1 and s[2] == 0
        open(os.path.join(tmpdir, "ok2"), "w").write("ok")
    dist.destroy_process_group()


def _bench_bookkeeping_worker(rank, world, port, tmpdir):
    # bench.py's own job functions (barrier, slowest rank's time, the shared-table
    # exchange, whole-job throughput) under gloo: the N > 1 branches the benchmark
    # runs on the GPUs, executed here on CPU tensors
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    import oracle
    hfz = importlib.import_module("huffman-avx512_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one
        return len(calls)

    steps, warmup = 5, 2
    elapsed, last = bench.time_steps(step, steps, warmup, world, False, cpu)
    assert last == steps + warmup and len(calls) == steps + warmup  # exactly K timed steps
    assert elapsed >= steps * 0.01 * world * 0.95                     # the slowest rank's time
    both = [None] * world
    dist.all_gather_object(both, elapsed)
    assert both[0] == both[1]                                         # the same number on every rank
    assert bench.job_value(world, 256.0, 2.0, 4) == world * 256.0 / 0.5
    assert bench.max_over_ranks(float(rank), world, cpu) == float(world - 1)

    # the shared-table exchange: per-rank counts, one of them beyond 32 bits
    h = torch.zeros(256, dtype=torch.int64)
    h[rank] = (1 << 33) + 5 * rank
    h[7] = 3
    h[200 + rank] = 1
    want_sum = np.zeros(256, dtype=np.uint64)
    for r in range(world):
        want_sum[r] += (1 << 33) + 5 * r
        want_sum[7] += 3
        want_sum[200 + r] += 1
    h32 = bench.job_histogram(hfz, h, world)
    got = h32.numpy().view(np.uint32)
    want = hfz.scale_histogram(want_sum)
    assert got.tolist() == want.tolist()
    assert int(want.astype(np.uint64).sum()) < (1 << 32) and want[7] >= 1 and want[201] == 1
    assert h[rank] == (1 << 33) + 5 * rank                            # the caller's tensor is untouched
    # every rank builds the same code from it (host builder here, device builder on the GPUs)
    rec = bytes(hfz.frame_code(got))
    recs = [None] * world
    dist.all_gather_object(recs, rec)
    assert recs[0] == recs[1]
    coding = oracle.make_coding(got)
    assert bytes(hfz.frame_code(got).hdr)[4 + bin(coding.len_mask).count("1"):][:coding.num_syms] == \
        coding.sorted_syms.tobytes()
    if rank == 0:
        open(os.path.join(tmpdir, "ok3"), "w").write("ok")
    dist.destroy_process_group()


def test_two_ranks_bench_bookkeeping(tmp_path):
    import torch.multiprocessing as mp
    port = 30700 + (os.getpid() % 500)
    mp.spawn(_bench_bookkeeping_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok3").exists()


def test_scale_histogram_keeps_sums_below_32_bits():
    # the code builders add node counts in 32 bits (as codec/huffman.cpp:364-416
    # does): after scaling, the Huffman code they build must be a true Huffman code
    # of the scaled counts, i.e. cost what a 64-bit construction costs
    import heapq
    import numpy as np
    hfz = importlib.import_module("huffman-avx512_amd")
    rng = np.random.default_rng(7)
    for case in range(20):
        nsym = int(rng.integers(2, 60))
        h = np.zeros(256, dtype=np.uint64)
        idx = rng.choice(256, size=nsym, replace=False)
        h[idx] = rng.integers(1, 1 << 36, size=nsym).astype(np.uint64)  # sums far beyond 2^32
        s = hfz.scale_histogram(h)
        assert int(s.astype(np.uint64).sum()) < (1 << 32)
        assert ((s > 0) == (h > 0)).all()
        import torch
        assert hfz.scale_histogram_dev(torch.from_numpy(h.astype(np.int64))).numpy().view(
            np.uint32).tolist() == s.tolist()
        t = hfz.build_code_table(s)
        lens = np.array(list(t.len), dtype=np.int64)
        cost = int((s.astype(np.int64) * lens).sum())
        heap = [int(c) for c in s if c]
        heapq.heapify(heap)
        best = 0
        while len(heap) > 1:
            a, b = heapq.heappop(heap), heapq.heappop(heap)
            best += a + b
            heapq.heappush(heap, a + b)
        if lens.max() < 12:  # no length limiting happened: the code is optimal
            assert cost == best, case
        else:
            assert cost >= best


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
