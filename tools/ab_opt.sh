#!/bin/bash
# GPU box: bench.py's 4K line under a list of --opt settings (launch-shape A/B without a rebuild).
# Usage: bash tools/ab_opt.sh <tag> "<opt list or empty>" ...   -> gpurun_out/<tag>_opt.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=$1; shift; mkdir -p $OUT
B="--batch 64 --steps 8 --warmup 2 --cpu-frames 0 --repeats 1"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-26s' % '$1', d['value'], 'frames/s ', ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for pass in 1 2; do
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python3 $R/bench.py $B $args 2>/dev/null | line "${o:-default}"
done
done
} 2>&1 | tee $OUT/${TAG}_opt.txt
