"""Dev tool (GPU box): frames/s of pythoncrt_amd.process_frames fed by an in-memory iterator of numpy frames and a writer that does nothing — the
ceiling of the Python-level drop-in for process_video's loop (ref:1037-1131), host side included (one memcpy per frame into the pinned batch,
PCIe both ways, the per-frame write_frame call).     python tools/process_frames_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pythoncrt_amd as pc

for (h, w, n) in ((1080, 1920, 480), (2160, 3840, 160)):
    rng = np.random.default_rng(0)
    base = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(8)]
    for name, kw in (("reference CLI defaults", {}), ("full chain: Gaussian bloom sigma 3 + warp 0.15", dict(fast_bloom=False, bloom_sigma=3.0, warp_strength=0.15, pixel_size=1, persistence=0.0))):
        pc.process_frames(iter(base), lambda a: None, w, h, 30, 8, noise_seed=1, **kw)                 # warm-up: ctx, tables, pinned slots
        t = time.perf_counter()
        k = pc.process_frames((base[i % 8] for i in range(n)), lambda a: None, w, h, 30, n, noise_seed=1, **kw)
        dt = time.perf_counter() - t
        print(f"{w}x{h} {name}: {k} frames in {dt:.3f} s = {k / dt:.0f} frames/s (set-up of the call included)", flush=True)
