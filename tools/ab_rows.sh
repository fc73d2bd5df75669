#!/bin/bash
# GPU box: WARP_ROWS = 2 / 4 on the other BASELINE configs.   -> gpurun_out/<tag>_rows.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=${1:-rows}; mkdir -p $OUT
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-30s' % '$1', d['value'], 'frames/s ', ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for i in 1 2; do
for c in 2 4 5; do
for rows in 2 4; do
python3 $R/bench.py --config $c --steps 4 --warmup 1 --cpu-frames 0 --repeats 1 --opt WARP_ROWS=$rows 2>/dev/null | line "config $c rows $rows"
done; done; done
} 2>&1 | tee $OUT/${TAG}_rows.txt
