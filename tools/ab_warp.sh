#!/bin/bash
# GPU box: k_warp_lean tile shapes — the in-tree library and build/ab/*.so (WL_WX builds) under WARP_ROWS = 2, 4, 8.
# Usage: bash tools/ab_warp.sh <tag> [passes]   -> gpurun_out/<tag>_warp.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=${1:-w}; N=${2:-2}; mkdir -p $OUT
B="--batch 64 --steps 8 --warmup 2 --cpu-frames 0 --repeats 1"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-30s' % '$1', d['value'], 'frames/s ', ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for i in $(seq $N); do
for rows in 2 4 8; do
python3 $R/bench.py $B --opt WARP_ROWS=$rows 2>/dev/null | line "wx1 rows $rows"
for L in $R/build/ab/*.so; do CRTFX_LIB=$L python3 $R/bench.py $B --opt WARP_ROWS=$rows 2>/dev/null | line "$(basename $L) rows $rows"; done
done
done
} 2>&1 | tee $OUT/${TAG}_warp.txt
