#!/bin/bash
# Dev tool (GPU box): A/B builds under build/ab/ on the configs k_phosphor_rr serves (1080p R = 4, 8K half frames).  tools/ab_rr.sh <variant|base> ...
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; cd $R
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$R/build/ab/lib_$v.so"
  for cfg in 2 5 4; do
    b=256; [ $cfg = 5 ] && b=16
    CRTFX_LIB=$lib timeout -k 10 150 python bench.py --config $cfg --steps 8 --warmup 2 --repeats 0 --cpu-frames 0 --batch $b > $OUT/abrr_${v}_c$cfg.json 2>/dev/null
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/abrr_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); k=r["roofline"]["kernels"]
        print(os.path.basename(f), r["value"], {n:v["avg_launch_ms"] for n,v in k.items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
