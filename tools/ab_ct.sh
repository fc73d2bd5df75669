#!/bin/bash
# GPU box: A/B of k_phosphor_ct (composite triad tables, centre samples from the frame) against k_phosphor_cc and against the builds
# under build/ab/*.so (python -c "from pythoncrt_amd import _lib; _lib.build(force=True, extra_flags=[...], out='build/ab/libcrtfx_X.so')").
# Usage: bash tools/ab_ct.sh <tag> [passes]  -> gpurun_out/<tag>_ab.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=${1:-ab}; N=${2:-2}; mkdir -p $OUT
B="--batch 64 --steps 8 --warmup 2 --cpu-frames 0 --repeats 1"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-22s' % '$1', d['value'], 'frames/s  repeats', d.get('repeat_values'), ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for i in $(seq $N); do
python3 $R/bench.py $B 2>/dev/null | line "ct(default)"
python3 $R/bench.py $B --opt NO_CT=1 2>/dev/null | line "cc(NO_CT)"
for L in $R/build/ab/*.so; do CRTFX_LIB=$L python3 $R/bench.py $B 2>/dev/null | line "$(basename $L)"; done
done
} 2>&1 | tee $OUT/${TAG}_ab.txt
