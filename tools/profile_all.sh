#!/bin/bash
# One profiling session on the GPU box: kernel times + traffic (tools/profile.sh),
# SQ counters (tools/pmc.sh) and the bench lines of the named configurations.
# usage: tools/profile_all.sh <tag>     results under gpurun_out/
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
bash tools/profile.sh "$TAG" || exit 1
bash tools/pmc.sh "$TAG" > "gpurun_out/pmc_$TAG.txt" 2>&1 || exit 2
timeout -k 10 200 python3 bench.py > "gpurun_out/${TAG}_bench.json" 2> "gpurun_out/${TAG}_bench.err" || exit 3
timeout -k 10 120 python3 bench.py --no-cpu-baseline --frame-kib 1024 > "gpurun_out/${TAG}_1mib_bench.json" 2>/dev/null || exit 4
timeout -k 10 120 python3 bench.py --no-cpu-baseline --frame-kib 256 > "gpurun_out/${TAG}_256k_bench.json" 2>/dev/null || exit 5
for C in 3 4 5; do
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --config $C > "gpurun_out/${TAG}_cfg${C}_bench.json" 2>/dev/null || exit 6
done
tail -n 1 "gpurun_out/${TAG}_bench.json"
