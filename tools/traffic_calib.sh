#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/probe/traffic_calib (known byte counts) on the GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/calib
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in 1024 2048; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$OUT/${C}_$W" -- \
        "$R/tools/probe/traffic_calib" $W > "$OUT/${C}_$W.log" 2>&1 || { tail -5 "$OUT/${C}_$W.log"; exit 2; }
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
for W in (1024, 2048):
    true = W * 64 * 2048
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%s_%d/*/*_counter_collection.csv" % (out, C, W)):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != C: continue
                name = re.sub(r"\(.*", "", r["Kernel_Name"])
                print("waves %d %-10s %-28s %10.1f KiB = %.3f of the %d KiB moved" % (
                    W, C, name, float(r["Counter_Value"]), float(r["Counter_Value"]) * 1024 / true, true >> 10))
PY
