#!/bin/bash
# SQ counter passes over bench.py (diagnostic).  usage: tools/pmc.sh <tag> [bench args]
TAG=${1:-pmc}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
SET1=${PMC_SET1:-"SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"}
SET2=${PMC_SET2:-"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"}
for SET in "$SET1" "$SET2"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $SET --output-format csv -d "$OUT/p$i" -- \
      python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-side "$@" > "$OUT/p$i.log" 2>&1 || { tail -5 "$OUT/p$i.log"; exit 2; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, re, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel(?:_wide)?)(?:<[^>]*>)?\(", r["Kernel_Name"])
        if m and "hfz::" in r["Kernel_Name"]:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
