#!/bin/bash
# PCIe-inclusive end-to-end rate of the CLI (raw rgb24 file in tmpfs -> /dev/null), start-up included and steady state:
#   bash tools/cli_throughput.sh [W H N_short N_long]      -> stdout (append to gpurun_out/<tag>_cli.txt yourself)
# Two clips of N_short and N_long frames per flag set; steady state = (N_long - N_short) / (t_long - t_short), which cancels the
# start-up (imports, ctx, table uploads, the first batch's unoverlapped legs).  The long run's --staging-report line says where the
# reader / GPU-feeding / writer threads spent their time.
W=${1:-3840}; H=${2:-2160}; NS=${3:-240}; NL=${4:-960}
F=/dev/shm/crtfx_in_${W}x${H}.rgb
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
with open("$F", "wb") as f:
    blk = rng.integers(0, 256, (8, $H, $W, 3), dtype=np.uint8)
    for i in range($NL // 8):
        f.write(np.roll(blk, i * 7, axis=2).tobytes())
PY
cd "$(dirname "$0")/.."
LOG=$(mktemp)
run() {   # frames flags... -> elapsed seconds on stdout, the CLI's stderr in $LOG
  local n=$1; shift
  head -c $((n * W * H * 3)) $F > ${F}.part
  python3 -m pythoncrt_amd.cli --input ${F}.part --output /dev/null --width $W --height $H --fps 30 --batch ${BATCH:-16} --noise-seed 1 --staging-report "$@" 2>$LOG >/dev/null
  sed -n 's/.*elapsed \([0-9.]*\)s.*/\1/p' $LOG
}
for flags in "--no-fast-bloom --bloom-sigma 3 --warp-strength 0.15 --pixel-size 1 --persistence 0" ""; do
  ts=$(run $NS $flags); tl=$(run $NL $flags)
  python3 - <<PY
ts, tl, ns, nl = float("$ts"), float("$tl"), $NS, $NL
print(f"${W}x${H}  flags: ${flags:-<reference defaults>}")
print(f"    {ns} frames {ts:.3f} s = {ns / ts:.0f} frames/s    {nl} frames {tl:.3f} s = {nl / tl:.0f} frames/s    steady state {(nl - ns) / (tl - ts):.0f} frames/s")
PY
  grep staging $LOG | sed 's/^/    /'
done
rm -f $F ${F}.part $LOG
