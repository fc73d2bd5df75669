#!/bin/bash
# GPU box: A/B of k_phosphor_ct (composite triad tables, 5 blocks per CU) against k_phosphor_cc and against its own 4-block build.
# Usage: bash tools/ab_ct.sh <tag>   -> gpurun_out/<tag>_ab.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=${1:-ab}; mkdir -p $OUT
B="--batch 64 --steps 8 --warmup 2 --cpu-frames 0 --repeats 1"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('$1', d['value'], 'frames/s  repeats', d.get('repeat_values'), ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for i in 1 2; do
python3 $R/bench.py $B | line "ct(default)"
python3 $R/bench.py $B --opt NO_CT=1 | line "cc(NO_CT)"
for L in $R/build/ab/*.so; do CRTFX_LIB=$L python3 $R/bench.py $B | line "$(basename $L)"; done
done
} 2>&1 | tee $OUT/${TAG}_ab.txt
