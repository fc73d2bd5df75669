#!/bin/bash
# GPU box: A/B of the CLI-default chain (bench.py --config 0) — k_point_fused_seq (default) against k_half_group + k_point_lean_seq (NO_FUSED_HALF=1),
# and the fused kernel at 16 wavefronts per block.  Interleaved repeats on one box.   bash tools/ab_fused.sh [tag]  -> gpurun_out/<tag>_ab_fused.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-r06}
F=$OUT/${TAG}_ab_fused.txt
: > $F
one() { python3 $R/bench.py --config 0 --cpu-frames 0 --repeats 1 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print('%-34s %9.0f frames/s  repeats %s  kernels %s  plan %s' % (sys.argv[1], d['value'], d['repeat_values'], {k: (round(v['avg_launch_ms'] * 1e3, 1), v['frames_per_launch']) for k, v in r['kernels'].items()}, r['plan'].get('point')))
" "$*" >> $F; }
for rep in 1 2 3; do
  one
  one --opt NO_FUSED_HALF=1
  one --opt POINT_TILES=6
  one --opt POINT_TILES=4
done
cat $F
