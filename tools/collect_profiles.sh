#!/bin/bash
# Run on the GPU box (gpurun): collects the rocprofv3 evidence bench.py's roofline object refers to.
# Usage: bash tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01_z}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --steps 10 --cpu-frames 0 > $OUT/${TAG}_trace.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --batch 4 --cpu-frames 0 --no-profile > $OUT/${TAG}_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $R/bench.py --steps 3 --warmup 1 --batch 4 --cpu-frames 0 --no-profile > $OUT/${TAG}_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${TAG}_sq -- python3 $R/bench.py --steps 3 --warmup 1 --batch 4 --cpu-frames 0 --no-profile > $OUT/${TAG}_sq.log 2>&1
cat $OUT/${TAG}_bench.json
