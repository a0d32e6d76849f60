#!/usr/bin/env python3
"""Condenses a tools/pmc.sh run (gpurun_out/pmc_<tag>.log, its last lines) into
profiles/<tag>_sq.json: per kernel the raw SQ counter means per launch and the
figures derived from them (instructions per symbol and per wave, issue activity
per wave, share of LDS-active cycles lost to bank conflicts).
usage: summarize_pmc.py <tag> <raw_bytes_per_launch>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, raw_bytes = sys.argv[1], float(sys.argv[2])
    out = {"tag": tag, "raw_bytes_per_launch": raw_bytes, "kernels": {}}
    for line in open(os.path.join(ROOT, "gpurun_out", "pmc_%s.log" % tag)):
        m = re.match(r"(\w+_kernel) (.*)", line.strip())
        if not m:
            continue
        c = dict((k, float(v)) for k, v in (kv.split("=") for kv in m.group(2).split()))
        wave_symbols = raw_bytes / 64.0  # one lane per stream: 64 symbols per wave-symbol
        d = {"counters": c}
        if c.get("SQ_WAVES"):
            d["per_wave_symbol"] = {k: round(c[n] / wave_symbols, 2) for k, n in (
                ("valu", "SQ_INSTS_VALU"), ("lds", "SQ_INSTS_LDS"), ("salu", "SQ_INSTS_SALU")) if n in c}
            if c.get("SQ_WAVE_CYCLES"):
                d["active_inst_any_over_wave_cycles"] = round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
                d["wait_any_over_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
            if c.get("SQ_LDS_IDX_ACTIVE"):
                d["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
        out["kernels"][m.group(1)] = d
    path = os.path.join(ROOT, "profiles", tag + "_sq.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    for k, d in out["kernels"].items():
        print(k, {x: d[x] for x in d if x != "counters"})


if __name__ == "__main__":
    main()
