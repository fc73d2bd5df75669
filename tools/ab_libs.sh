#!/bin/bash
# GPU box: the in-tree library against every build/ab/*.so (dev A/B builds, _lib.build_variant) on one bench.py workload.
# Usage: bash tools/ab_libs.sh <tag> [bench.py flags]   -> gpurun_out/<tag>_libs.txt (appended)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=$1; shift; mkdir -p $OUT
B="--steps 6 --warmup 2 --cpu-frames 0 --repeats 1 $*"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-34s' % '$1', d['value'], 'frames/s ', ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for i in 1 2 3; do
python3 $R/bench.py $B 2>/dev/null | line "in-tree [$*]"
for L in $R/build/ab/*.so; do CRTFX_LIB=$L python3 $R/bench.py $B 2>/dev/null | line "$(basename $L) [$*]"; done
done
} 2>&1 | tee -a $OUT/${TAG}_libs.txt
