#!/bin/bash
# GPU box: memory-pipeline counters of the chain's kernels (TA busy, vector-memory instructions, TA FIFO back-pressure, TA <-> TC stalls) for one option set.
# Usage: bash tools/pmc_vmem.sh <tag> [bench flags, e.g. --config 5 --opt NO_CT=1]   -> gpurun_out/<tag>_vmem.json
#
# Every pass is its own rocprofv3 run with AT MOST TWO counters of the TA block (round 5: six TA / TCP counters in one pass made
# rocprofiler_create_counter_config fail with "error code 38: Request exceeds the capabilities of the hardware to collect" and the profiled
# process die with SIGABRT — gpurun_out/r05_c5_tc.log — while this script carried on and wrote a JSON without them).  A pass that exits non-zero,
# aborts inside the tool or leaves no counter CSV ends the script non-zero NAMING THE PASS, and no further GPU pass is started; the aggregator
# refuses to write the JSON unless every counter of every pass is present for every crtfx kernel it saw.
set -euo pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-vmem}; shift || true
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
PASSES=(
  "ta|TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
  "vm|SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  "tc1|TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  "tc2|TCP_PENDING_STALL_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum"
  "tc3|TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum"
)
for spec in "${PASSES[@]}"; do
  name=${spec%%|*}; counters=${spec#*|}
  log=$OUT/${TAG}_$name.log
  rm -rf "$OUT/${TAG}_$name"
  rc=0
  # the program itself directly after `--` (no env / bash -c hop under the profiler)
  timeout -k 10 200 rocprofv3 --pmc $counters --output-format csv -d "$OUT/${TAG}_$name" -- \
      python3 "$R/bench.py" --steps 2 --warmup 1 --batch 8 --repeats 0 --cpu-frames 0 --no-profile "$@" > "$log" 2>&1 || rc=$?
  if [ $rc -ne 0 ] || grep -q "caught signal\|failed with error code" "$log"; then
    echo "pmc_vmem: pass '$name' ($counters) failed, exit $rc — see $log; no further pass started" >&2
    grep -m 3 "error code\|caught signal" "$log" >&2 || true
    exit 1
  fi
  if ! find "$OUT/${TAG}_$name" -name '*counter_collection.csv' | grep -q .; then
    echo "pmc_vmem: pass '$name' wrote no counter_collection.csv — see $log" >&2
    exit 1
  fi
  echo "pmc_vmem: pass $name ok"
done
python3 "$R/tools/pmc_vmem_aggregate.py" "$OUT" "$TAG" "${PASSES[@]}"
