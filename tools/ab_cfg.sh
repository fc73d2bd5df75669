#!/bin/bash
# GPU box: bench.py on one config with option sets.  Usage: bash tools/ab_cfg.sh <tag> <config> "<opts A>" "<opts B>" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=$1; CFG=$2; shift; shift; mkdir -p $OUT
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-28s' % '$1', d['value'], 'frames/s  repeats', d.get('repeat_values'), ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for o in "$@"; do python3 $R/bench.py --config $CFG --batch 256 --steps 8 --warmup 2 --cpu-frames 0 --repeats 1 $o 2>/dev/null | line "cfg$CFG $o"; done
} 2>&1 | tee -a $OUT/${TAG}_cfg.txt
