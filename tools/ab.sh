#!/bin/bash
# GPU box: one bench.py workload under a list of --opt settings, on the in-tree library and on every dev build under build/ab/*.so
# (_lib.build_variant), interleaved and repeated so that every comparison is made inside one gpurun call (boxes differ by +-5 %).
# Usage: bash tools/ab.sh <tag> "<bench.py flags>" ["<opt list>" ...]      e.g.  bash tools/ab.sh w "--batch 64" "" "WARP_ROWS=2" "WARP_ROWS=1 GROUP=1"
#   -> gpurun_out/<tag>_ab.txt (appended); PASSES=n repeats the list n times (default 2)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; TAG=$1; FL=$2; shift; shift; mkdir -p $OUT
[ $# -eq 0 ] && set -- ""
B="--steps 6 --warmup 2 --cpu-frames 0 --repeats 1 $FL"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); k=r.get('kernels',{})
print('%-48s' % '$1', d['value'], 'frames/s ', ' '.join(f\"{n} {v['avg_launch_ms']*1e3:.1f}us/{v['frames_per_launch']:.0f}f\" for n,v in k.items()))"; }
{
for i in $(seq ${PASSES:-2}); do
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python3 $R/bench.py $B $args 2>/dev/null | line "in-tree [$FL] ${o:-default}"
  for L in $R/build/ab/*.so; do if [ -e "$L" ]; then CRTFX_LIB=$L python3 $R/bench.py $B $args 2>/dev/null | line "$(basename $L) [$FL] ${o:-default}"; fi; done
done; done
} 2>&1 | tee -a $OUT/${TAG}_ab.txt
