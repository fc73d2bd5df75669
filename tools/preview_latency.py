"""Per-tick latency of the preview path (apply_crt_effect as the reference GUI calls it, ref:1810-1852) at 1080p:
numpy frame in / numpy frame + state out (PCIe both ways) and device tensors in / out."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pythoncrt_amd as pc
h, w = 1080, 1920
rng = np.random.default_rng(0)
frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
tm, vg = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
state = None
def tick(i, state):
    return pc.apply_crt_effect(frame, 0.6, tm, 2.2, False, 1, 1.2, 0.25, 0.0, 1.5, vg, 0.2, state, 2.0, i * 1.0, True, 2, time_sec=i / 30.0)
for i in range(3):
    out, state = tick(i, state)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for i in range(n):
    out, state = tick(i, state)
torch.cuda.synchronize()
print(f"numpy in/out: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per tick (1080p, CLI-default-like preview settings)")
ft = torch.from_numpy(frame).cuda(); st = None
def tick_t(i, st):
    return pc.apply_crt_effect(ft, 0.6, tm, 2.2, False, 1, 1.2, 0.25, 0.0, 1.5, vg, 0.2, st, 2.0, i * 1.0, True, 2, time_sec=i / 30.0)
for i in range(3):
    o, st = tick_t(i, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    o, st = tick_t(i, st)
torch.cuda.synchronize()
print(f"tensor in/out: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per tick")
