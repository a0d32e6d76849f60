#!/bin/bash
# Frame-size sweep of bench.py (per-kernel times), for tuning.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for F in "$@"; do
  timeout -k 10 120 python3 -u "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined --frame-kib $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('frame_kib=%s value=%.0f ms=%.3f ratio=%.4f '%(d['config']['frame_kib'], d['value'], d['ms_per_step'], d['config']['compressed_over_raw']) + ' '.join('%s=%.3f'%(n,k[n]['ms'] or 0) for n in k))
" || exit 1
done
