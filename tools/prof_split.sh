#!/bin/bash
# rocprofv3 kernel stats of the split-bloom path at a few sigmas (4K full chain): tools/prof_split.sh <tag> "<sigmas>" [bench_sigma flags]
tag=${1:-split}; sigmas=${2:-"10.5 42"}
cd /tmp && export TMPDIR=/tmp
for s in $sigmas; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_s$s -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_sigma.py --sigmas $s --steps 3 "${@:3}" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_s$s.log 2>&1
  echo "== $tag sigma $s: $(grep '^sigma' $GRAFT_REPO_ROOT/gpurun_out/${tag}_s$s.log)"
  python3 - <<PY
import csv
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/${tag}_s$s/p_kernel_stats.csv")):
    if 'crtfx' in r['Name']: print(f"  {r['Name'][:60]:60s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1000:8.1f} us")
PY
done
