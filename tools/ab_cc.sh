#!/bin/bash
# Dev tool (GPU box): phase stamps of k_phosphor_cc / k_phosphor_rr and the timing-experiment builds under build/ab/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
B="--steps 8 --warmup 2 --repeats 0 --cpu-frames 0 --tables-outside --batch 64"
for v in "$@"; do
  case $v in
    stamp) CRTFX_LIB=$R/build/ab/lib_stamp.so timeout -k 10 120 python tools/phase_profile.py 3 > $OUT/ab_stamp_cc.txt 2>&1; CRTFX_LIB=$R/build/ab/lib_stamp.so timeout -k 10 120 python tools/phase_profile.py 3 NO_CC=1 > $OUT/ab_stamp_rr.txt 2>&1;;
    base) timeout -k 10 120 python bench.py $B > $OUT/ab_base.json 2>$OUT/ab_base.err; timeout -k 10 120 python bench.py $B --opt NO_CC=1 > $OUT/ab_base_rr.json 2>>$OUT/ab_base.err;;
    rrnostore) CRTFX_LIB=$R/build/ab/lib_nostore.so timeout -k 10 120 python bench.py $B --opt NO_CC=1 > $OUT/ab_rrnostore.json 2>$OUT/ab_rrnostore.err;;
    g1) timeout -k 10 120 python bench.py $B --opt GROUP=1 > $OUT/ab_g1.json 2>$OUT/ab_g1.err; timeout -k 10 120 python bench.py $B --opt GROUP=1 --opt NO_CC=1 > $OUT/ab_g1_rr.json 2>>$OUT/ab_g1.err;;
    cfg5) timeout -k 10 150 python bench.py --config 5 --steps 6 --warmup 2 --repeats 0 --cpu-frames 0 --tables-outside --batch 16 > $OUT/ab_cfg5.json 2>$OUT/ab_cfg5.err; timeout -k 10 150 python bench.py --config 5 --steps 6 --warmup 2 --repeats 0 --cpu-frames 0 --tables-outside --batch 16 --opt NO_CC=1 > $OUT/ab_cfg5_rr.json 2>>$OUT/ab_cfg5.err;;
    cfg2) timeout -k 10 150 python bench.py --config 2 --steps 8 --warmup 2 --repeats 0 --cpu-frames 0 --tables-outside --batch 128 > $OUT/ab_cfg2.json 2>$OUT/ab_cfg2.err; timeout -k 10 150 python bench.py --config 2 --steps 8 --warmup 2 --repeats 0 --cpu-frames 0 --tables-outside --batch 128 --opt NO_CC=1 > $OUT/ab_cfg2_rr.json 2>>$OUT/ab_cfg2.err;;
    seg*) timeout -k 10 120 python bench.py $B --opt SEG_ROWS=${v#seg} > $OUT/ab_$v.json 2>$OUT/ab_$v.err;;
    wbw) timeout -k 10 120 ./build/ubench/write_bw > $OUT/r02_write_bw.txt 2>&1; cat $OUT/r02_write_bw.txt;;
    *) CRTFX_LIB=$R/build/ab/lib_$v.so timeout -k 10 120 python bench.py $B > $OUT/ab_$v.json 2>$OUT/ab_$v.err;;
  esac
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        k=r["roofline"]["kernels"]
        print(os.path.basename(f), r["value"], {n:v["avg_launch_ms"] for n,v in k.items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
cat $OUT/ab_stamp_*.txt 2>/dev/null
