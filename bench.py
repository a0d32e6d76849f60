#!/usr/bin/env python3
"""Benchmark of the K-stream Huffman hot path on MI355X.

One step = encode one batch of synthetic frames and decode it again, both
through the C ABI (hfz_compress_batch_dev / hfz_decompress_batch_dev), with the
raw input already resident in HBM.  At N GPUs every rank runs its own batch of
the same size (independent frames, no data-path collective): weak scaling.

Prints one JSON line (rank 0).  `value` = raw MiB that went through encode AND
decode per second, whole job: N * total / (time per step).  MiB = 2^20 bytes of
RAW data, the reference's convention (codec/huffman_benchmark.cpp:67,78).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def main():
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from rocprofv3 PMC passes, if collected")
    presets = {2: {}, 3: dict(streams=32, total_mib=1024, source="enwik", frame_kib=512),
               4: dict(total_mib=8192, frame_kib=1024),
               5: dict(total_mib=32768, frame_kib=1024, shared_table=True)}
    pre, _ = ap.parse_known_args()
    if pre.config:
        ap.set_defaults(**presets[pre.config])
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    hfz = importlib.import_module("huffman-avx512_amd")
    hfz.lib()  # fails loudly if the HIP library is missing

    K = args.streams
    total = args.total_mib << 20
    frame_bytes = args.frame_kib << 10
    cdf = hfz.distributions.SOURCES[args.source]()
    raw = hfz.generate(cdf, 1000 + rank, total, device=dev)
    codec = hfz.BatchCodec(K, total, frame_bytes, device=dev)
    out = torch.empty(total, dtype=torch.uint8, device=dev)

    L = hfz.lib()
    code = torch.zeros(792, dtype=torch.uint8, device=dev)

    def shared_code():
        # one pass over the batch leaves per-stream counts in the workspace and the
        # 256 byte counts here; those are summed over ranks: the job's one exchange,
        # 2 KiB per GPU
        st = torch.cuda.current_stream(dev).cuda_stream
        h = codec.count(raw).clone()
        if world > 1:
            dist.all_reduce(h)
        # counts beyond 32 bits are scaled down, every occurring symbol stays nonzero
        shift = torch.clamp(torch.floor(torch.log2(h.max().clamp(min=1).double())) - 31, min=0).long()
        h = torch.where(h > 0, torch.clamp(h >> shift, min=1), h)
        h32 = (h & 0xffffffff).to(torch.int32)  # bit pattern of the unsigned count
        rc = L.hfz_build_codes_dev(h32.data_ptr(), 1, code.data_ptr(), st)
        assert rc == 0, rc
        return code

    def step():
        if args.shared_table:
            batch = codec.compress(raw, code=shared_code(), counted=True)
            codec.decompress(batch, out, shared=True)
        else:
            batch = codec.compress(raw)
            codec.decompress(batch, out)
        return batch

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    hfz.timing_enable(max(1, min(args.time_every, args.steps)))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch = step()
    barrier()
    t1 = time.perf_counter()
    kt = hfz.timing_read()
    hfz.timing_enable(False)

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed
    assert torch.equal(out, raw), "round trip mismatch"
    assert int(codec.status.abs().sum().item()) == 0 and int(codec.dstatus.abs().sum().item()) == 0
    comp_bytes = batch.compressed_bytes()

    ms_per_step = 1e3 * elapsed / args.steps
    mib = total / float(1 << 20)
    value = world * mib / (elapsed / args.steps)

    # per-kernel roofline: algorithmic HBM bytes per launch / mean launch time
    N, C = float(total), float(comp_bytes)
    alg_bytes = {"hist": N, "sizes": N, "encode": N + C, "decode": C + N}
    kernels = {}
    for name in ("hist", "table", "sizes", "place", "encode", "decode"):
        ms = kt[name + "_ms"]
        ent = {"ms": round(ms, 4)}
        # sizes_ms is the gap between two adjacent events when the input pass already
        # gave the stream sizes: no kernel ran, so no rate
        if name in alg_bytes and ms > (0.02 if name == "sizes" else 0):
            ent["GBps"] = round(alg_bytes[name] / (ms * 1e-3) / 1e9, 1)
        kernels[name] = ent
    dom = max(("hist", "sizes", "encode", "decode"), key=lambda k: kt[k + "_ms"])
    achieved = alg_bytes[dom] / (kt[dom + "_ms"] * 1e-3) / 1e9 if kt[dom + "_ms"] > 0 else 0.0
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
    enc_ms = sum(kt[k + "_ms"] for k in ("hist", "table", "sizes", "place", "encode"))
    roofline = {"bound": "hbm", "kernel": dom + "_kernel", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": int(alg_bytes[dom]),
                "launch_ms": round(kt[dom + "_ms"], 4),
                "launches_timed": int(kt["decompress_calls"] if dom == "decode" else kt["compress_calls"])}

    result = {
        "metric": "MiB/s encode+decode on 256-sym biased input; achieved HBM GB/s vs peak",
        "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d-stream encode+decode, %d MiB %s synthetic per GPU, "
                               "%d KiB frames, %s" % (
                                   K, args.total_mib,
                                   "biased-geometric" if args.source == "biased" else "enwik8-like",
                                   args.frame_kib,
                                   "one shared code (device histogram summed over ranks), tables "
                                   "in LDS" if args.shared_table else "per-frame tables in LDS"),
                   "shared_table": bool(args.shared_table),
                   "source": args.source,
                   "streams": K, "raw_mib_per_gpu": args.total_mib, "frame_kib": args.frame_kib,
                   "frames_per_gpu": codec.n_frames, "compressed_over_raw": round(C / N, 4),
                   "layout": "packed"},
        "encode_MiBps_kernels": round(mib / (enc_ms * 1e-3), 1) if enc_ms > 0 else None,
        "decode_MiBps_kernels": round(mib / (kt["decode_ms"] * 1e-3), 1) if kt["decode_ms"] > 0 else None,
        "kernels": kernels,
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The oracle (scalar C restatement of the reference, test infrastructure) on the
        # host cores of this box, whole-call compress + decompress per frame like the
        # reference's benchmark; a bounded sample of the same batch (~10 s per leg).
        import oracle
        sample = min(total, 32 << 20)
        host = raw[:sample].cpu().numpy()
        smib1 = sample / float(1 << 20)
        try:
            cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
        except ValueError:
            cores = min(16, os.cpu_count() or 1)
        cores = max(1, min(cores, 64))

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
        result["cpu_baseline"] = {
            "value": multi["value"], "unit": "MiB/s", "cores": cores, "kind": "port",
            "encode_MiBps": multi["encode_MiBps"], "decode_MiBps": multi["decode_MiBps"],
            "sample": "first %d MiB of rank 0's batch (%d frames of %d KiB, K=%d), %d repetitions of "
                      "whole-call compress+decompress with the scalar C restatement of the "
                      "reference, frames spread over %d threads" % (
                          sample >> 20, sample // frame_bytes, args.frame_kib, K, multi["reps"], cores),
            "single_thread": {"value": single["value"], "encode_MiBps": single["encode_MiBps"],
                              "decode_MiBps": single["decode_MiBps"], "reps": single["reps"]},
            "k4_100KiB_single_thread": {"value": small["value"], "encode_MiBps": small["encode_MiBps"],
                                        "decode_MiBps": small["decode_MiBps"], "reps": small["reps"]},
            "k32_single_thread": {"value": k32["value"], "encode_MiBps": k32["encode_MiBps"],
                                  "decode_MiBps": k32["decode_MiBps"], "reps": k32["reps"]}}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
