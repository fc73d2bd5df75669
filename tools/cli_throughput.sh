#!/bin/bash
# PCIe-inclusive end-to-end rate of the CLI (raw rgb24 file in tmpfs -> /dev/null): bash tools/cli_throughput.sh [W H N]
W=${1:-3840}; H=${2:-2160}; N=${3:-96}
F=/dev/shm/crtfx_in_${W}x${H}.rgb
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
with open("$F", "wb") as f:
    blk = rng.integers(0, 256, (8, $H, $W, 3), dtype=np.uint8)
    for i in range($N // 8):
        f.write(np.roll(blk, i * 7, axis=2).tobytes())
PY
cd "$(dirname "$0")/.."
for flags in "--no-fast-bloom --bloom-sigma 3 --warp-strength 0.15 --pixel-size 1 --persistence 0" ""; do
  echo "flags: ${flags:-<reference defaults>}"
  python3 -m pythoncrt_amd.cli --input $F --output /dev/null --width $W --height $H --fps 30 --batch 16 --noise-seed 1 $flags
done
rm -f $F
