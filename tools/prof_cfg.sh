#!/bin/bash
# rocprofv3 kernel stats of one bench config: tools/prof_cfg.sh <tag> <config> [bench flags]
tag=${1:-cfg}; cfg=${2:-0}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag} -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 5 --warmup 2 --repeats 0 --cpu-frames 0 --no-profile "${@:3}" > $GRAFT_REPO_ROOT/gpurun_out/${tag}.log 2>&1
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/${tag}.log | cut -c1-200
python3 - <<PY
import csv
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/${tag}/p_kernel_stats.csv")):
    if 'crtfx' in r['Name']: print(f"  {r['Name'][:90]:90s} n={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.1f} us  {r['Percentage']}%")
PY
