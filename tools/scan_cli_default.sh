# Frames/s of the reference CLI default chain at 1080p with one knob changed at a time (which kernel build each lands on).
B="fast_bloom=1 pixel_size=2 persistence=0.2 warp_strength=0"
for S in "" "fast_bloom=0" "fast_bloom=0 pixel_size=1" "fast_bloom=0 pixel_size=1 warp_strength=0.15" "grain_size=2" "scanline_angle=5" "triad_preserve_luma=1" "noise_strength=0" "gamma=1.8" "saturation=1.3" "glitch_amp_px=9 glitch_height_frac=0.3" "persistence=0" "pixel_size=3" "warp_strength=0.15" "flicker_strength=0.4 flicker_hz=9" "bloom_threshold=0.3"; do
  echo "== $S"; timeout -k 10 200 python tools/bench_sigma.py --sigmas 1.2 --h 1080 --w 1920 --batch 32 --set $B $S 2>&1 | tail -1
done
for O in before after; do echo "== overlay $O"; timeout -k 10 200 python tools/bench_sigma.py --sigmas 1.2 --h 1080 --w 1920 --batch 32 --overlay $O --set $B 2>&1 | tail -1; done
