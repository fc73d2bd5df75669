#!/usr/bin/env python3
"""GPU box: frames/s of the render loop (FramePipeline.run, frames resident in HBM) for the reference CLI's default settings and for settings ONE KNOB
away from them — which kernel builds a user lands on when they touch a flag, and what it costs (crtfx_last_plan beside each rate).

    python tools/bench_cli_variants.py [--height 1080 --width 1920 --frames 512 --reps 5]  -> one line per variant"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pythoncrt_amd.pipeline import FramePipeline, RenderSettings  # noqa: E402

VARIANTS = [
    ("defaults", {}),
    ("--brightness 0.05", dict(brightness=0.05)),
    ("--contrast 1.1", dict(contrast=1.1)),
    ("--saturation 1.2", dict(saturation=1.2)),
    ("--gamma 1.2", dict(gamma=1.2)),
    ("--temperature 0.2", dict(temperature=0.2)),
    ("--bloom-threshold 0.3", dict(bloom_threshold=0.3)),
    ("--pixel-size 1", dict(pixel_size=1)),
    ("--pixel-size 3", dict(pixel_size=3)),
    ("--persistence 0", dict(persistence=0.0)),
    ("--noise 0", dict(noise_strength=0.0)),
    ("--vignette 0", dict(vignette_strength=0.0)),
    ("--triad 0", dict(triad_strength=0.0)),
    ("--scanlines 0", dict(scanline_strength=0.0)),
    ("--triad-preserve-luma", dict(triad_preserve_luma=True)),
    ("--grain-size 2", dict(grain_size=2)),
    ("--flicker 0.1 @ 50 Hz", dict(flicker_strength=0.1, flicker_hz=50.0)),
    ("--scanline-angle 10", dict(scanline_angle=10.0)),
    ("--scanline-thickness 2", dict(scanline_thickness=2.0)),
    ("--warp-strength 0.15", dict(warp_strength=0.15)),
    ("--glitch 4 px / 0.2", dict(glitch_amp_px=4, glitch_height_frac=0.2)),
    ("--no-fast-bloom (sigma 1.2)", dict(fast_bloom=False)),
    ("--bloom-strength 0", dict(bloom_strength=0.0)),
    ("--brightness 0.05 --saturation 1.2", dict(brightness=0.05, saturation=1.2)),
    ("--noise 0 --vignette 0", dict(noise_strength=0.0, vignette_strength=0.0)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    frames = torch.randint(0, 256, (a.frames, a.height, a.width, 3), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty_like(frames)
    print(f"{a.width}x{a.height}, {a.frames} frames per run, best of {a.reps}")
    for name, kw in VARIANTS:
        if a.only and a.only not in name:
            continue
        rs = RenderSettings(**kw)
        pipe = FramePipeline(dev, a.height, a.width, rs, fps=30.0, noise_seed=1)
        _, st = pipe.run(frames, first_index=0, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for r in range(a.reps):
            t0 = time.perf_counter()
            _, st = pipe.run(frames, first_index=(r + 1) * a.frames, state=st, out=out)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        pipe.profile(1)                       # HIP events on every launch of one more run: kernel time per launch, by class
        pipe.run(frames, first_index=(a.reps + 1) * a.frames, state=st, out=out)
        torch.cuda.synchronize()
        kt = {k: f"{v[0] * 1e3:.1f} us / {v[2] / max(1, v[1]):.0f} fr" for k, v in pipe.profile_read().items() if v[1]}
        pipe.profile(False)
        plan = pipe.plan()
        builds = " + ".join(str(plan[k]) for k in ("blur", "half", "phosphor", "point", "warp") if plan.get(k))
        print(f"{name:30s} {a.frames / best:10.0f} frames/s   group {plan.get('group')}   {builds}   {kt}", flush=True)
        del pipe


if __name__ == "__main__":
    main()
