#!/bin/bash
# PCIe-inclusive end-to-end rate of the CLI's raw-rgb24 edge, start-up included and steady state:
#   bash tools/cli_throughput.sh [W H N_short N_long [DIR ...]]      -> stdout (append to gpurun_out/<tag>_cli.txt yourself)
# For every DIR (default /dev/shm and /tmp: tmpfs and the box's disk-backed filesystem behave very differently, profiles/r05_hostreg_probe*.txt)
# two clips of N_short and N_long frames per case; steady state = (N_long - N_short) / (t_long - t_short), which cancels the start-up (imports,
# ctx, table uploads, the first batch's unoverlapped legs).  Cases (4K headline flags unless FLAGS is set):
#   file -> /dev/null      --io staged | mapped | auto         the input leg alone (what rounds 1-4 called "the CLI rate")
#   file -> new file       --io staged | mapped                both legs; a NEW output file is bound by the kernel's page allocation
#   file -> existing file  --io mapped                         both legs with the output's pages already in the page cache
#   pipe -> pipe           cat file | cli - - | cat > /dev/null      the mode INTEGRATION.md recommends between two ffmpeg processes
# The long run's --staging-report lines say where the reader / GPU-feeding / writer threads spent their time and which path each leg took.
W=${1:-3840}; H=${2:-2160}; NS=${3:-240}; NL=${4:-960}
shift; shift; shift; shift
DIRS=${@:-/dev/shm /tmp}
FLAGS=${FLAGS:---no-fast-bloom --bloom-sigma 3 --warp-strength 0.15 --pixel-size 1 --persistence 0}
cd "$(dirname "$0")/.."
LOG=$(mktemp)
for D in $DIRS; do
F=$D/crtfx_in_${W}x${H}.rgb
NEED=$((NL * W * H * 3 / 1024 * 33 / 10))       # input + its cut copy + one output file, in KiB, + 10 %
FREE=$(df --output=avail -k $D | tail -1)
if [ "$FREE" -lt "$NEED" ]; then echo "== $D: $((FREE / 1048576)) GiB free, $((NEED / 1048576)) needed for ${NL} frames of ${W}x${H}: skipped"; continue; fi
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
with open("$F", "wb") as f:
    blk = rng.integers(0, 256, (8, $H, $W, 3), dtype=np.uint8)
    for i in range($NL // 8):
        f.write(np.roll(blk, i * 7, axis=2).tobytes())
PY
FS=$(df --output=fstype $D | tail -1)
echo "== ${W}x${H}, input and output files in $D ($FS); flags: $FLAGS"
run() {   # frames input output io -> elapsed seconds on stdout, the CLI's stderr in $LOG
  local n=$1 in=$2 out=$3 io=$4
  head -c $((n * W * H * 3)) $F > ${F}.part
  if [ "$in" = "-" ]; then
    cat ${F}.part | python3 -m pythoncrt_amd.cli --input - --output - --width $W --height $H --fps 30 --batch ${BATCH:-16} --noise-seed 1 --staging-report $FLAGS 2>$LOG | cat > /dev/null
  else
    python3 -m pythoncrt_amd.cli --input ${F}.part --output $out --width $W --height $H --fps 30 --batch ${BATCH:-16} --noise-seed 1 --staging-report --io $io $FLAGS 2>$LOG >/dev/null
  fi
  grep -q elapsed $LOG || { echo "      (run failed:)" >&2; tail -5 $LOG | sed 's/^/      /' >&2; }
  sed -n 's/.*elapsed \([0-9.]*\)s.*/\1/p' $LOG
}
case_() {  # label input output io [keep-output-between-runs]
  local label=$1 in=$2 out=$3 io=$4 keep=$5
  if [ -n "$ONLY" ] && ! echo "$label" | grep -Eq "$ONLY"; then return; fi      # ONLY=<regex>: a subset of the cases
  [ -z "$keep" ] && [ "$out" != "/dev/null" ] && [ "$out" != "-" ] && rm -f $out
  ts=$(run $NS $in $out $io)
  [ -z "$keep" ] && [ "$out" != "/dev/null" ] && [ "$out" != "-" ] && rm -f $out
  tl=$(run $NL $in $out $io)
  python3 - <<PY
ts, tl, ns, nl = float("$ts" or "nan"), float("$tl" or "nan"), $NS, $NL
print(f"  $label: {ns} frames {ts:.3f} s, {nl} frames {tl:.3f} s -> steady state {(nl - ns) / (tl - ts):.0f} frames/s")
PY
  grep "staging: pipeline" $LOG | sed 's/^staging: /      in-process: /'
  grep staging $LOG | sed 's/^/      /'
}
for io in staged mapped auto; do case_ "file -> /dev/null, --io $io" file /dev/null $io; done
if [ -n "$KNOBS" ]; then      # A/B of the staged reader's knobs (KNOBS=1)
  CRTFX_IO_DONTNEED=0 case_ "file -> /dev/null, staged, no MADV_DONTNEED behind the copies" file /dev/null staged
  CRTFX_IO_DONTNEED=slice case_ "file -> /dev/null, staged, MADV_DONTNEED per slice on the copy threads (round 4)" file /dev/null staged
  CRTFX_IO_THREADS=32 case_ "file -> /dev/null, staged, 32 I/O threads" file /dev/null staged
  CRTFX_IO_THREADS=8 case_ "file -> /dev/null, staged, 8 I/O threads" file /dev/null staged
  BATCH=32 case_ "file -> /dev/null, staged, batches of 32 frames" file /dev/null staged
fi
for io in staged mapped; do case_ "file -> new file, --io $io" file ${F}.out $io; done
# the output file of the run before is still there, full size: its pages are in the page cache
head -c $((NL * W * H * 3)) $F > ${F}.out
case_ "file -> existing file (pages resident), --io mapped" file ${F}.out mapped keep
case_ "pipe -> pipe (cat | cli | cat > /dev/null), 1 MiB pipe buffers (F_SETPIPE_SZ)" - - staged
CRTFX_PIPE_SIZE=65536 case_ "pipe -> pipe, the default 64 KiB pipe buffers" - - staged
rm -f $F ${F}.part ${F}.out
done
rm -f $LOG
