#!/bin/bash
# GPU box: memory-pipeline counters of the chain's kernels (TA busy, vector-memory instructions, TA FIFO back-pressure) for one option set.
# Usage: bash tools/pmc_vmem.sh <tag> [bench flags, e.g. --opt NO_CT=1]   -> gpurun_out/<tag>_vmem.json
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-vmem}; shift
cd /tmp; export TMPDIR=/tmp
run() { timeout -k 10 200 rocprofv3 --pmc $2 --output-format csv -d $OUT/${TAG}_$1 -- python3 $R/bench.py --steps 2 --warmup 1 --batch 8 --repeats 0 --cpu-frames 0 --no-profile "${@:3}" > $OUT/${TAG}_$1.log 2>&1; }
run ta "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "$@"
run vm "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "$@"
run tc "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" "$@"
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${TAG}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "crtfx" in k:
            agg[k.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$OUT/${TAG}_vmem.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k)
    for c in sorted(d): print(f"   {c:36s} {d[c]:16.0f}")
PY
