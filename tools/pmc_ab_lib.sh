#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for which in intree forcegrade; do
  if [ $which = forcegrade ]; then export CRTFX_LIB=$R/build/ab/libcrtfx_forcegrade.so; else unset CRTFX_LIB; fi
  for pass in "a|SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "b|SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_IFETCH"; do
    n=${pass%%|*}; c=${pass#*|}
    rm -rf $OUT/fg_${which}_$n
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/fg_${which}_$n -- python3 $R/bench.py --config 0 --steps 2 --warmup 1 --batch 64 --repeats 0 --cpu-frames 0 --no-profile > $OUT/fg_${which}_$n.log 2>&1 || { echo "pass $which $n failed"; tail -5 $OUT/fg_${which}_$n.log; exit 1; }
  done
done
python3 - <<PY
import csv, glob, collections
for which in ("intree","forcegrade"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/fg_%s_*/**/*counter_collection.csv" % which, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_point_fused_seq" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(which, {k: round(sum(v)/len(v)/1e6,3) for k,v in sorted(agg.items())})
PY
