#!/bin/bash
# GPU box, ONE call: everything bench.py's roofline object cites, for the headline and the per-config lines, from the same box —
#   per config: tools/collect_profiles.sh (bench line, rocprofv3 --kernel-trace --stats, the --pmc passes, FETCH_SIZE calibration)
#               -> tools/summarise_profiles.py (profiles/traffic.json, valu.json, <tag>_{kernel_stats.csv,pmc.json,fetch_calibration.txt})
#               -> bench.py --config N once more, so that the committed line carries the counters of its own sources and box
#   then the files to commit are gathered under gpurun_out/profiles_out/ (gpurun merges gpurun_out/ back; copy them into profiles/).
# Usage: bash tools/evidence_round.sh r05_z        (configs 3 = headline tag, then 0 2 4 5 as <tag>_cN)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r05_z}; OUT=$R/gpurun_out
mkdir -p $OUT/profiles_out
cd $R
for c in 3 0 2 4 5; do
  t=$TAG; [ $c != 3 ] && t=${TAG}_c$c
  bash tools/collect_profiles.sh $t --config $c > $OUT/${t}_collect.log 2>&1
  python3 tools/summarise_profiles.py $t config$c > $OUT/${t}_summarise.log 2>&1
  timeout -k 10 300 python3 bench.py --config $c > $OUT/${t}_bench_final.json 2> $OUT/${t}_bench_final.err
  tail -1 $OUT/${t}_bench_final.json > $OUT/profiles_out/${t}_bench.json
  cp profiles/${t}_kernel_stats.csv profiles/${t}_pmc.json profiles/${t}_fetch_calibration.txt $OUT/profiles_out/ 2>/dev/null
  echo "config $c done: $(python3 -c "import json; d=json.load(open('$OUT/profiles_out/${t}_bench.json')); r=d['roofline']; print(d['value'], r['bound'], r['frac'], r['traffic_source'].get('source_hash'), {k: round(v['avg_launch_ms'] * 1e3, 1) for k, v in r['kernels'].items()})")"
done
cp profiles/traffic.json profiles/valu.json $OUT/profiles_out/
