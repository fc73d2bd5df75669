#!/usr/bin/env python3
"""Frames/s of the 4K (or --h/--w) full chain as a function of the bloom sigma (radius = round(3 sigma)):
one fused build per radius up to 30, then the split bloom path (any radius)."""
import argparse, dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pythoncrt_amd.pipeline import FramePipeline, baseline_config

ap = argparse.ArgumentParser()
ap.add_argument("--sigmas", type=float, nargs="+", default=[1.2, 3.0, 4.0, 4.5, 6.5, 8.3, 10.0, 10.5, 12.0, 16.0, 17.0, 21.0, 22.0, 32.0, 33.0, 42.0])
ap.add_argument("--h", type=int, default=2160); ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--half", action="store_true", help="float16 frames (BASELINE configs[4]'s pixel format)")
ap.add_argument("--overlay", choices=["none", "before", "after"], default="none", help="a text overlay blended before / after the effects")
ap.add_argument("--set", nargs="*", default=[], metavar="field=value", help="RenderSettings overrides, e.g. scanline_angle=12 grain_size=2")
ap.add_argument("--opt", nargs="*", default=[], metavar="NAME=VALUE", help="crtfx_set_option switches, e.g. SPLIT_FROM=129")
a = ap.parse_args()
from pythoncrt_amd import effects
effects.DEBUG_OPTIONS = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.opt}
dev = torch.device("cuda", 0)
frames = torch.randint(0, 256, (a.batch, a.h, a.w, 3), dtype=torch.uint8, device=dev)
if a.half:
    frames = frames.to(torch.float16)
for s in a.sigmas:
    over = {}
    for kv in a.set:
        k, v = kv.split("=")
        cur = getattr(baseline_config(3)[0], k)
        over[k] = type(cur)(v) if not isinstance(cur, bool) else v.lower() in ("1", "true")
    rs = dataclasses.replace(baseline_config(3)[0], bloom_sigma=s, **over)
    ov = None
    if a.overlay != "none":
        from pythoncrt_amd.text import make_text_overlay_rgba
        ov = make_text_overlay_rgba(a.w, a.h, "PythonCRT on MI355X", "", 96, "#FFCC00", (64, 64))
    pipe = FramePipeline(dev, a.h, a.w, rs, fps=30.0, noise_seed=1, text_overlay_rgba=ov, text_overlay_after=(a.overlay == "after"), dtype=frames.dtype)
    out = torch.empty_like(frames)
    for _ in range(2):
        pipe.run(frames, out=out)
    torch.cuda.synchronize()
    ts = []
    for i in range(a.steps):      # per-frame tables (and device-generated scanline planes) are part of the timed loop here
        t0 = time.perf_counter()
        pipe.run(frames, first_index=i * a.batch, out=out)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]      # median step: a first-touch / clock-ramp outlier does not decide a row
    plan = pipe.plan().get("phosphor", pipe.plan().get("blur", ""))
    print(f"sigma {s:5.2f}  radius {max(1, int(round(s * 3)) * 2 + 1) // 2:3d}  {a.batch / dt:9.1f} frames/s  {dt / a.batch * 1e6:8.1f} us/frame  (max step {max(ts) / a.batch * 1e6:8.1f})  {plan}", flush=True)
