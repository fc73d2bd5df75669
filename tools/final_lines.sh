#!/bin/bash
# GPU box: the committed bench line of a round and the N > 1 rehearsal (several gloo ranks on the box's ONE GPU: the line end to end, not a scaling result;
# the box admits six GPU processes and the launcher is one of them, hence five ranks).   bash tools/final_lines.sh   -> gpurun_out/r04_z_bench.json, r04_five_rank_rehearsal.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
python3 $R/bench.py > $OUT/r04_z_bench.json 2> $OUT/r04_z_bench.err
echo "--- 5 ranks, config 3" > $OUT/r04_five_rank_rehearsal.txt
CRTFX_DIST_BACKEND=gloo timeout -k 10 400 python3 $R/bench.py --gpus 5 --config 3 --batch 64 --steps 4 --warmup 1 --repeats 0 --cpu-frames 0 >> $OUT/r04_five_rank_rehearsal.txt 2>$OUT/r04_five_c3.err
echo "--- 5 ranks, config 4" >> $OUT/r04_five_rank_rehearsal.txt
CRTFX_DIST_BACKEND=gloo timeout -k 10 400 python3 $R/bench.py --gpus 5 --config 4 --batch 256 --steps 3 --warmup 1 --repeats 0 --cpu-frames 0 >> $OUT/r04_five_rank_rehearsal.txt 2>$OUT/r04_five_c4.err
tail -c 400 $OUT/r04_five_c3.err $OUT/r04_five_c4.err
wc -c $OUT/r04_five_rank_rehearsal.txt
