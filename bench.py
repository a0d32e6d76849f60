#!/usr/bin/env python3
"""Benchmark of the K-stream Huffman hot path on MI355X.

One step = encode one batch of synthetic frames and decode it again, both
through the C ABI (hfz_compress_batch_dev / hfz_decompress_batch_dev), with the
raw input already resident in HBM.  At N GPUs every rank runs its own batch of
the same size (independent frames, no data-path collective): weak scaling.

Prints one JSON line (rank 0).  `value` = raw MiB that went through encode AND
decode per second, whole job: N * total / (time per step).  MiB = 2^20 bytes of
RAW data, the reference's convention (codec/huffman_benchmark.cpp:67,78).

The job bookkeeping (barrier, max-over-ranks time, the one exchange of the
shared-table mode) lives in functions of its own so that the world_size-2 gloo
test drives the same code on CPU (tests/test_multirank.py).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
METRIC = "MiB/s encode+decode on 256-sym biased input; achieved HBM GB/s vs peak"

# --force-dist: a one-rank job takes every branch a multi-rank job takes (process
# group, barriers, all-reduces), so those branches can be run on a single GPU.
FORCE_DIST = False


def _dist_on(world):
    return world > 1 or FORCE_DIST


# ---- job bookkeeping (runs under gloo on CPU as well) -------------------------

def job_barrier(world, cuda):
    """Device idle on every rank, then all ranks together, then idle again."""
    import torch
    if cuda:
        torch.cuda.synchronize()
    if _dist_on(world):
        import torch.distributed as dist
        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


def max_over_ranks(seconds, world, device):
    """The slowest rank's time: what the whole job took."""
    if not _dist_on(world):
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_histogram(hfz, h, world):
    """The shared-table mode's one exchange: the ranks' 256 byte counts (int64
    tensor, 2 KiB per GPU) are summed, then scaled to what the 32-bit code
    builders take.  Returns the int32 tensor of uint32 bit patterns, identical on
    every rank."""
    h = h.clone()
    if _dist_on(world):
        import torch.distributed as dist
        dist.all_reduce(h)
    return hfz.scale_histogram_dev(h)


def time_steps(step, steps, warmup, world, cuda, device):
    """`warmup` untimed steps, then exactly `steps` steps between two job barriers;
    returns (seconds of the slowest rank, what the last step returned)."""
    last = None
    for _ in range(warmup):
        last = step()
    job_barrier(world, cuda)
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    job_barrier(world, cuda)
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed, world, device), last


def job_value(world, mib_per_rank, elapsed, steps):
    """Whole-job raw MiB per second (every rank processed mib_per_rank per step)."""
    return world * mib_per_rank / (elapsed / steps)


# ---- the benchmark ---------------------------------------------------------------

def kernel_table(kt, alg_bytes, shared_table):
    """Per-kernel mean launch times with the algorithmic rate of those that moved
    data.  A kernel the library did not run in this mode gets ms = None, no rate."""
    ran = {"hist": (kt["count_calls"] > 0) if shared_table else kt["hist_ms"] > 0.002,
           "table": kt["table_ms"] > 0.002, "sizes": kt["sizes_ms"] > 0.02,
           "place": True, "encode": True, "decode": True}
    ms_of = dict((n, kt[n + "_ms"]) for n in ran)
    if shared_table:
        ms_of["hist"] = kt["count_ms"]  # the input pass is hfz_count_batch_dev's
    out = {}
    for name in ("hist", "table", "sizes", "place", "encode", "decode"):
        if not ran[name]:
            out[name] = {"ms": None}
            continue
        ent = {"ms": round(ms_of[name], 4)}
        if name in alg_bytes and ms_of[name] > 0:
            ent["GBps"] = round(alg_bytes[name] / (ms_of[name] * 1e-3) / 1e9, 1)
        out[name] = ent
    return out, ms_of, ran


class Workload:
    """One benchmark workload on one device: buffers, the step, its measurement."""

    def __init__(self, hfz, dev, rank, world, K, total_mib, frame_kib, source, shared_table,
                 in_flight=1):
        import torch
        self.torch, self.hfz, self.L, self.dev, self.rank, self.world = torch, hfz, hfz.lib(), dev, rank, world
        self.K, self.total_mib, self.frame_kib = K, total_mib, frame_kib
        self.source, self.shared_table = source, bool(shared_table)
        self.total = total_mib << 20
        self.frame_bytes = frame_kib << 10
        self.cdf = hfz.distributions.SOURCES[source]()
        self.slots = [self._slot(i, in_flight > 1) for i in range(max(1, in_flight))]
        torch.cuda.synchronize(dev)
        self.turn = 0

    def _slot(self, i, own_stream):
        torch, hfz, dev = self.torch, self.hfz, self.dev

        class Slot:
            """One batch with everything a step on it touches."""
        sl = Slot()
        sl.raw = hfz.generate(self.cdf, 1000 + self.rank + 7919 * i, self.total, device=dev)
        sl.codec = hfz.BatchCodec(self.K, self.total, self.frame_bytes, device=dev)
        sl.out = torch.empty(self.total, dtype=torch.uint8, device=dev)
        sl.code = torch.zeros(792, dtype=torch.uint8, device=dev)
        sl.stream = torch.cuda.Stream(dev) if own_stream else None
        return sl

    def add_slot(self):
        """A second batch, and a stream for each: the two-batches-in-flight figure."""
        torch = self.torch
        second = self._slot(len(self.slots), True)
        for sl in self.slots:
            if sl.stream is None:
                sl.stream = torch.cuda.Stream(self.dev)
        self.slots.append(second)
        torch.cuda.synchronize(self.dev)

    def _shared_code(self, sl):
        # one pass over the batch leaves per-stream counts in the workspace and the
        # 256 byte counts in a device tensor; those are summed over ranks (the job's
        # one exchange) and the code is built on the device from the sum
        h32 = job_histogram(self.hfz, sl.codec.count(sl.raw), self.world)
        rc = self.L.hfz_build_codes_dev(h32.data_ptr(), 1, sl.code.data_ptr(),
                                        self.torch.cuda.current_stream(self.dev).cuda_stream)
        assert rc == 0, rc
        return sl.code

    def _step_on(self, sl):
        if self.shared_table:
            batch = sl.codec.compress(sl.raw, code=self._shared_code(sl), counted=True)
            sl.codec.decompress(batch, sl.out, shared=True)
        else:
            batch = sl.codec.compress(sl.raw)
            sl.codec.decompress(batch, sl.out)
        return batch

    def step(self):
        sl = self.slots[self.turn % len(self.slots)]
        self.turn += 1
        if sl.stream is None:
            return self._step_on(sl)
        with self.torch.cuda.stream(sl.stream):
            return self._step_on(sl)

    def check(self, what=""):
        """Correctness of what was timed."""
        torch = self.torch
        for sl in self.slots:
            assert torch.equal(sl.out, sl.raw), "round trip mismatch" + what
            assert int(sl.codec.status.abs().sum().item()) == 0
            assert int(sl.codec.dstatus.abs().sum().item()) == 0

    def timed(self, steps, warmup, time_every):
        """(seconds of the slowest rank, last batch, per-kernel timing dict)."""
        hfz = self.hfz
        for _ in range(warmup):
            self.step()
        job_barrier(self.world, True)
        hfz.timing_enable(max(1, min(time_every, steps)) if time_every else 0)
        elapsed, batch = time_steps(self.step, steps, 0, self.world, True, self.dev)
        kt = hfz.timing_read()
        hfz.timing_enable(False)
        self.check()
        return elapsed, batch, kt

    def describe(self):
        return "%d-stream encode+decode, %d MiB %s synthetic per GPU, %d KiB frames, %s" % (
            self.K, self.total_mib,
            "biased-geometric" if self.source == "biased" else "enwik8-like", self.frame_kib,
            "one shared code (device histogram summed over ranks), tables in LDS"
            if self.shared_table else "per-frame tables in LDS")


def side_measurement(hfz, dev, rank, K, total_mib, frame_kib, source, steps, warmup, time_every):
    """A further workload measured after the headline one (one rank): whole-step
    throughput and the decoder's algorithmic rate against the HBM peak."""
    import torch
    w = Workload(hfz, dev, rank, 1, K, total_mib, frame_kib, source, False)
    elapsed, batch, kt = w.timed(steps, warmup, time_every)
    N, C = float(w.total), float(batch.compressed_bytes())
    mib = w.total / float(1 << 20)
    ent = {"workload": w.describe(), "value": round(job_value(1, mib, elapsed, steps), 1),
           "unit": "MiB/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 4),
           "frames": w.slots[0].codec.n_frames, "compressed_over_raw": round(C / N, 4)}
    enc_ms = kt["hist_ms"] + kt["table_ms"] + kt["sizes_ms"] + kt["place_ms"] + kt["encode_ms"]
    if kt["decode_ms"] > 0:
        gbps = (C + N) / (kt["decode_ms"] * 1e-3) / 1e9
        ent["decode"] = {"ms": round(kt["decode_ms"], 4), "GBps": round(gbps, 1),
                         "frac": round(gbps / HBM_PEAK_GBPS, 4)}
    if kt["encode_ms"] > 0:
        gbps = (C + N) / (kt["encode_ms"] * 1e-3) / 1e9
        ent["encode"] = {"ms": round(kt["encode_ms"], 4), "GBps": round(gbps, 1),
                         "frac": round(gbps / HBM_PEAK_GBPS, 4),
                         "all_encoder_kernels_ms": round(enc_ms, 4)}
    del w
    torch.cuda.empty_cache()
    return ent


def main():
    global FORCE_DIST
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64, help="K, streams per frame")
    ap.add_argument("--total-mib", type=int, default=256, help="raw MiB per GPU per step")
    ap.add_argument("--frame-kib", type=int, default=128, help="raw KiB per frame")
    ap.add_argument("--source", choices=("biased", "enwik"), default="biased",
                    help="biased: P(i)=0.8^i*0.2 (reference benchmark); enwik: wiki-text-like table")
    ap.add_argument("--shared-table", action="store_true",
                    help="one code for the whole job: device histogram (summed over ranks), "
                         "code built on the device, all frames coded and decoded with it")
    ap.add_argument("--config", type=int, choices=(2, 3, 4, 5), default=None,
                    help="preset of the named configuration: 2 = K64 256 MiB biased (default), "
                         "3 = K32 1 GiB enwik-like in 512 KiB frames, 4 = K64 8 GiB/GPU in 1 MiB frames, "
                         "5 = K64 32 GiB/GPU in 1 MiB frames with a shared table; "
                         "explicit flags given after it win")
    ap.add_argument("--time-every", type=int, default=4,
                    help="HIP events around each kernel on every n-th step of the timed region "
                         "(events between kernels cost about 5 %% of a step, so they are sampled)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="independent batches in flight, each on a HIP stream of its own, steps "
                         "taking them in turn (1 = every step behind the one before it, which is "
                         "what the per-kernel times and the roofline are quoted on; with 2 the "
                         "serial code construction of one batch runs beside the memory-bound "
                         "passes of the other)")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra measurement with two batches in flight")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the extra workloads measured after the headline one "
                         "(32-stream text-like 1 GiB; the headline batch in 1 MiB frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decoder-table-bits", type=int, default=0,
                    help="diagnostic: force the decoder's table build (11, 12, -1 = table-free)")
    ap.add_argument("--decoder-store", type=int, default=0,
                    help="diagnostic: force the decoder's store mode (1 = every lane its own blocks, 2 = lane pairs)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank branches (process group over RCCL, barriers, "
                         "all-reduces) even with one rank, so that they can be run on one GPU")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from rocprofv3 PMC passes, if collected")
    presets = {2: {}, 3: dict(streams=32, total_mib=1024, source="enwik", frame_kib=512),
               4: dict(total_mib=8192, frame_kib=1024),
               5: dict(total_mib=32768, frame_kib=1024, shared_table=True)}
    pre, _ = ap.parse_known_args()
    if pre.config:
        ap.set_defaults(**presets[pre.config])
    args = ap.parse_args()
    FORCE_DIST = bool(args.force_dist)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" %
                         (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if _dist_on(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    hfz = importlib.import_module("huffman-avx512_amd")
    hfz.lib()  # fails loudly if the HIP library is missing
    if args.decoder_table_bits:
        hfz.set_decoder_table_bits(args.decoder_table_bits)
    if args.decoder_store:
        hfz.set_decoder_store(args.decoder_store)

    K = args.streams
    total = args.total_mib << 20
    frame_bytes = args.frame_kib << 10
    work = Workload(hfz, dev, rank, world, K, args.total_mib, args.frame_kib, args.source,
                    args.shared_table, args.in_flight)
    n_slots = len(work.slots)
    raw, codec = work.slots[0].raw, work.slots[0].codec
    elapsed, batch, kt = work.timed(args.steps, args.warmup, args.time_every)
    comp_bytes = batch.compressed_bytes()

    ms_per_step = 1e3 * elapsed / args.steps
    mib = total / float(1 << 20)
    value = job_value(world, mib, elapsed, args.steps)

    # per-kernel roofline: algorithmic HBM bytes per launch / mean launch time
    N, C = float(total), float(comp_bytes)
    alg_bytes = {"hist": N, "sizes": N, "encode": N + C, "decode": C + N}
    kernels, ms_of, ran = kernel_table(kt, alg_bytes, args.shared_table)
    dom = max((k for k in ("hist", "sizes", "encode", "decode") if ran[k]), key=lambda k: ms_of[k])
    achieved = alg_bytes[dom] / (ms_of[dom] * 1e-3) / 1e9 if ms_of[dom] > 0 else 0.0
    traffic = None
    try:  # PMC passes are separate rocprofv3 runs; use them only for the same workload
        with open(args.traffic_json) as f:
            prof = json.load(f)
        pc = (prof.get("bench_line") or {}).get("config", {})
        same = all(pc.get(k) == v for k, v in (("streams", K), ("raw_mib_per_gpu", args.total_mib),
                                               ("frame_kib", args.frame_kib),
                                               ("source", args.source),
                                               ("shared_table", bool(args.shared_table))))
        if same:
            traffic = prof.get(dom + "_kernel")
    except (OSError, ValueError):
        pass
    enc_ms = sum(ms_of[k] for k in ("hist", "table", "sizes", "place", "encode") if ran[k])
    launches = {"decode": kt["decompress_calls"],
                "hist": kt["count_calls"] if args.shared_table else kt["compress_calls"]}
    roofline = {"bound": "hbm", "kernel": dom + "_kernel", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": int(alg_bytes[dom]),
                "launch_ms": round(ms_of[dom], 4),
                "launches_timed": int(launches.get(dom, kt["compress_calls"]))}

    result = {
        "metric": METRIC,
        "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": work.describe(),
                   "shared_table": bool(args.shared_table),
                   "source": args.source,
                   "streams": K, "raw_mib_per_gpu": args.total_mib, "frame_kib": args.frame_kib,
                   "frames_per_gpu": codec.n_frames, "compressed_over_raw": round(C / N, 4),
                   "layout": "packed", "batches_in_flight": n_slots},
        # sums of kernel times, the input pass of the shared-table mode included
        "encode_MiBps_kernels": round(mib / (enc_ms * 1e-3), 1) if enc_ms > 0 else None,
        "decode_MiBps_kernels": round(mib / (kt["decode_ms"] * 1e-3), 1) if kt["decode_ms"] > 0 else None,
        "kernels": kernels,
        "roofline": roofline,
    }
    if FORCE_DIST:
        result["forced_dist"] = True

    # The same steps with two batches in flight (a caller with a queue of batches):
    # an extra figure beside `value`, measured after it, on one GPU and for batches
    # small enough that a second set of buffers is no burden.
    if world == 1 and n_slots == 1 and not args.no_pipelined and args.total_mib <= 2048:
        try:
            work.add_slot()
            p_elapsed, _ = time_steps(work.step, args.steps, 2, world, True, dev)
            work.check(" (two batches in flight)")
            result["pipelined"] = {"batches_in_flight": 2,
                                   "value": round(job_value(world, mib, p_elapsed, args.steps), 1),
                                   "unit": "MiB/s", "ms_per_step": round(1e3 * p_elapsed / args.steps, 4),
                                   "note": "independent batches taken in turn on two HIP streams; "
                                           "kernels of different batches share the chip, so no "
                                           "per-kernel times are quoted for it"}
        except Exception as e:  # e.g. no room for the second set of buffers: `value` stands
            result["pipelined"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    # Two more workloads beside the headline one, each a short run of its own after
    # it (one rank, default workload only): the 32-stream format on 1 GiB of
    # text-like data (BASELINE config[2]) and the headline batch cut into 1 MiB
    # frames (the frame size config[3] names), where a batch has 256 frames.
    default_line = (K == 64 and args.total_mib == 256 and args.frame_kib == 128 and
                    args.source == "biased" and not args.shared_table and args.in_flight == 1)
    if world == 1 and default_line and not args.no_side:
        host_raw = raw[:min(total, 32 << 20)].cpu() if not args.no_cpu_baseline else None
        del work, raw, codec, batch
        torch.cuda.empty_cache()
        for name, kw in (("k32_enwik", dict(K=32, total_mib=1024, frame_kib=512, source="enwik")),
                         ("frames_1mib", dict(K=64, total_mib=256, frame_kib=1024, source="biased"))):
            try:
                result[name] = side_measurement(hfz, dev, rank, steps=10, warmup=2,
                                                time_every=args.time_every, **kw)
            except Exception as e:  # the headline figures above stand whatever happens here
                result[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        raw = host_raw

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(raw, total, K, frame_bytes, args.frame_kib)
        except Exception as e:  # the GPU figures are not lost to a failing host-side leg
            result["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if _dist_on(world):
        dist.destroy_process_group()


def cpu_baseline(raw, total, K, frame_bytes, frame_kib):
    """The oracle (scalar C restatement of the reference, test infrastructure) on the
    host cores of this box, whole-call compress + decompress per frame like the
    reference's benchmark; a bounded sample of the same batch (~10 s per leg)."""
    import oracle
    sample = min(total, 32 << 20)
    host = raw[:sample].cpu().numpy()
    # threads used: the box's share for one GPU (OMP_NUM_THREADS there), not the
    # machine's core count, which host_cores reports beside it
    try:
        cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    except ValueError:
        cores = min(16, os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = None

    def leg(K_, data, fb, threads, seconds):
        tc, td, _ = oracle.time_roundtrip(K_, data, fb, 1, threads)
        reps = int(max(1, min(2000, round(seconds / max(tc + td, 1e-4)))))
        tc, td, _ = oracle.time_roundtrip(K_, data, fb, reps, threads)
        m = reps * len(data) / float(1 << 20)
        return {"value": round(m / (tc + td), 1), "encode_MiBps": round(m / tc, 1),
                "decode_MiBps": round(m / td, 1), "reps": reps}

    multi = leg(K, host, frame_bytes, cores, 8.0)
    single = leg(K, host, frame_bytes, 1, 8.0)
    # the reference's own CPU-runnable case: K = 4, one 100 KiB biased frame
    small = leg(4, host[:100 << 10], 100 << 10, 1, 2.0)
    k32 = leg(32, host, frame_bytes, 1, 3.0)  # the reference's widest scalar instantiation
    return {
        "value": multi["value"], "unit": "MiB/s", "cores": cores, "kind": "port",
        "host_cores": os.cpu_count(), "host_cores_usable": usable,
        "encode_MiBps": multi["encode_MiBps"], "decode_MiBps": multi["decode_MiBps"],
        "sample": "first %d MiB of rank 0's batch (%d frames of %d KiB, K=%d), %d repetitions of "
                  "whole-call compress+decompress with the scalar C restatement of the "
                  "reference, frames spread over %d threads" % (
                      sample >> 20, sample // frame_bytes, frame_kib, K, multi["reps"], cores),
        "single_thread": {"value": single["value"], "encode_MiBps": single["encode_MiBps"],
                          "decode_MiBps": single["decode_MiBps"], "reps": single["reps"]},
        "k4_100KiB_single_thread": {"value": small["value"], "encode_MiBps": small["encode_MiBps"],
                                    "decode_MiBps": small["decode_MiBps"], "reps": small["reps"]},
        "k32_single_thread": {"value": k32["value"], "encode_MiBps": k32["encode_MiBps"],
                              "decode_MiBps": k32["decode_MiBps"], "reps": k32["reps"]}}


if __name__ == "__main__":
    main()
