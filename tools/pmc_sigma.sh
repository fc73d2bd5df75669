#!/bin/bash
# GPU box: SQ counters of the split-bloom kernels at one sigma.  Usage: bash tools/pmc_sigma.sh <tag> <sigma> [bench_sigma flags...]
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-pmcs}; SIG=${2:-42}; shift; shift
cd /tmp; export TMPDIR=/tmp
run() { timeout -k 10 200 rocprofv3 --pmc $2 --output-format csv -d $OUT/${TAG}_$1 -- python3 $R/tools/bench_sigma.py --sigmas $SIG --steps 1 --batch 4 "${@:3}" > $OUT/${TAG}_$1.log 2>&1; }
run a "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "$@"
run b "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVE_CYCLES" "$@"
run e "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "$@"
# FETCH_SIZE costs 3 of the 4 TCC counter slots and WRITE_SIZE 2 (MI355X_MICROARCH.md, rocprofv3 PMC slots): one pass each —
# asked for together rocprofv3 aborts with "Request exceeds the capabilities of the hardware to collect" (error 38)
run f "FETCH_SIZE GRBM_GUI_ACTIVE" "$@"
run w "WRITE_SIZE" "$@"
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${TAG}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "crtfx" in k:
            agg[k.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$OUT/${TAG}_pmc_detail.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k)
    for c in sorted(d): print(f"   {c:28s} {d[c]:16.0f}")
PY
