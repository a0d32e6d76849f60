#!/bin/bash
# Profiles bench.py on the GPU box: per-kernel times (rocprofv3 --kernel-trace
# --stats) and, in separate passes, the HBM traffic counters.  Results land in
# gpurun_out/prof_<tag>/; tools/summarize_prof.py turns them into profiles/*.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-side "$@" > "$OUT/stats.log" 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- \
      python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-side "$@" > "$OUT/pmc_$C.log" 2>&1 || exit 2
done
find "$OUT" -name "*.csv" | head -20
tail -1 "$OUT/stats.log"
