#!/usr/bin/env python3
"""Condenses a tools/profile.sh run (gpurun_out/prof_<tag>/) into profiles/:
  profiles/<tag>_kernel_stats.csv  per-kernel rocprofv3 --stats rows (hfz kernels)
  profiles/<tag>_traffic.json      per-launch HBM bytes from the PMC passes
  profiles/traffic.json            the same, where bench.py looks for it

HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: rocprofv3
reports both counters in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte
requests of a wide coalesced read stream at 64 bytes each, so it is doubled
(MI355X microarchitecture guide, HBM section).  WRITE_SIZE is exact for
16-byte-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(\w+_kernel(?:_wide)?)(?:<[^>]*>)?\(", name)
    return m.group(1) if m else None


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
    rows = []
    with open(stats) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            if k and "hfz::" in r["Name"]:
                rows.append({"kernel": k, "calls": r["Calls"], "avg_us": "%.2f" % (float(r["AverageNs"]) / 1e3),
                             "min_us": "%.2f" % (float(r["MinNs"]) / 1e3), "max_us": "%.2f" % (float(r["MaxNs"]) / 1e3),
                             "pct": r["Percentage"]})
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, "pmc_" + counter, "*", "*_counter_collection.csv"))
        if not files:
            continue
        acc = {}
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if k and "hfz::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    acc.setdefault(k, []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            per.setdefault(k, {})[counter + "_KiB_per_launch"] = sum(v) / len(v)
    traffic = {}
    for k, d in per.items():
        fetch = d.get("FETCH_SIZE_KiB_per_launch")
        write = d.get("WRITE_SIZE_KiB_per_launch")
        if fetch is not None and write is not None:
            traffic[k] = int(2 * fetch * 1024 + write * 1024)
    log = os.path.join(src, "stats.log")
    bench = None
    for line in open(log):
        if line.startswith("{"):
            bench = json.loads(line)
    out = {"tag": tag, "raw_counters": per, "bench_line": bench, **traffic}
    for name in (tag + "_traffic.json", "traffic.json"):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(out, f, indent=1)
    for r in rows:
        print(r)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
