#!/bin/bash
# Run on the GPU box (gpurun): collects the rocprofv3 evidence bench.py's roofline object refers to.
# Usage: bash tools/collect_profiles.sh <tag> [bench flags, e.g. --config 2]    -> gpurun_out/<tag>_*   (then tools/summarise_profiles.py <tag> [configN] here)
# After a source change the bench line taken HERE still carries the previous build's traffic.json (nulled as stale): once
# summarise_profiles.py has written the new one, take `python bench.py > gpurun_out/<tag>_bench.json` again and copy it to profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02_z}
shift
X="$@"
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC passes: few steps, no event timing — but a batch far beyond the 256 MB Infinity Cache (default 256 frames: 6.4 GB in + 6.4 GB out at 4K), so
# that the input and output frames come from / go to HBM as in the bench's 1920-frame steps (rounds 1-3 measured at batch 8, where everything
# stayed cache-resident).  The batch is recorded in gpurun_out/<tag>_pmc_batch.txt for summarise_profiles.py.
PMC_BATCH=${PMC_BATCH:-256}
echo $PMC_BATCH > $OUT/${TAG}_pmc_batch.txt
PB="--steps 2 --warmup 1 --batch $PMC_BATCH --repeats 0 --cpu-frames 0 --no-profile"
timeout -k 10 300 python3 $R/bench.py $X > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py $X --steps 10 --repeats 0 --cpu-frames 0 > $OUT/${TAG}_trace.log 2>&1
rm -f $OUT/${TAG}_trace/*/*kernel_trace.csv      # the per-dispatch trace (tens of MB) is not needed: the stats CSV is what gets committed
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $R/bench.py $X $PB > $OUT/${TAG}_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $R/bench.py $X $PB > $OUT/${TAG}_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${TAG}_sq -- python3 $R/bench.py $X $PB > $OUT/${TAG}_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_sq2 -- python3 $R/bench.py $X $PB > $OUT/${TAG}_sq2.log 2>&1
# the raw counter CSVs of a 256-frame batch are tens of MB (gpurun brings back at most 64 MiB): keep the per-kernel means, drop the rows
for part in fetch write sq sq2; do
python3 - "$OUT/${TAG}_$part" <<'PY'
import collections, csv, glob, json, os, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    os.remove(f)
json.dump({k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in agg.items()}, open(d + "_agg.json", "w"))
PY
done
# FETCH_SIZE on a known, cold byte count in this code's own load shapes (tools/ubench/fetch_calib.hip)
if [ -x $R/build/ubench/fetch_calib ]; then
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calib -- $R/build/ubench/fetch_calib > $OUT/${TAG}_calib.log 2>&1
fi
cat $OUT/${TAG}_bench.json
