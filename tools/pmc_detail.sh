#!/bin/bash
# GPU box: extra --pmc passes behind DESIGN.md's instruction-mix / LDS paragraphs.  Usage: bash tools/pmc_detail.sh <tag> [bench flags...]
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-pmcx}; shift
cd /tmp; export TMPDIR=/tmp
run() { timeout -k 10 200 rocprofv3 --pmc $2 --output-format csv -d $OUT/${TAG}_$1 -- python3 $R/bench.py --steps 2 --warmup 1 --batch 8 --repeats 0 --cpu-frames 0 --no-profile "${@:3}" > $OUT/${TAG}_$1.log 2>&1; }
run a "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "$@"
run b "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVE_CYCLES" "$@"
run e "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "$@"
if [ "$PMC_FULL" = 1 ]; then
run c "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" "$@"
run d "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "$@"
fi
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
