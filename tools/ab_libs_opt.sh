#!/bin/bash
# GPU box: in-tree library and build/ab/*.so, each under a list of --opt settings.  Usage: bash tools/ab_libs_opt.sh <tag> "<bench flags>" "<opts>" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=$1; FL=$2; shift; shift; mkdir -p $OUT
B="--steps 6 --warmup 2 --cpu-frames 0 --repeats 1 $FL"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-44s' % '$1', d['value'], 'frames/s ', ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for i in 1 2; do
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python3 $R/bench.py $B $args 2>/dev/null | line "in-tree [$FL] ${o:-default}"
  for L in $R/build/ab/*.so; do CRTFX_LIB=$L python3 $R/bench.py $B $args 2>/dev/null | line "$(basename $L) [$FL] ${o:-default}"; done
done; done
} 2>&1 | tee -a $OUT/${TAG}_lo.txt
