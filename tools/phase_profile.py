#!/usr/bin/env python3
"""Dev tool (GPU box): per-phase cycle shares of k_phosphor_rr from a CRTFX_STAMP diagnostic build.

    python -c "from pythoncrt_amd import _lib; _lib.build(force=True, extra_flags=['-DCRTFX_STAMP'], out='build/stamp/lib_stamp.so')"
    python tools/phase_profile.py [config] [NO_CC=1 ...]

Slots: 0 A(grade->LDS) 1 barrier 2 B(H-pass) 3 barrier 4 C1(V-pass) 5 barrier 6 C2(masks+store).
k_phosphor_cc / k_phosphor_ct (waves 0-2 consumers, wave 3 helper): 4 = phase 1 up to and including the V pass (helper: vignette tile),
0 = the rest of phase 1 (A items, prefetch), 1 = wait at the first barrier, 6 = the tail C2 (helper: grain tile), 2 = H pass, 3 = wait at the
second barrier."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CRTFX_LIB", os.path.join(ROOT, "build", "stamp", "lib_stamp.so"))
import torch
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
n_words = 4096 * 4 * 8
dbg = torch.zeros(n_words, dtype=torch.int64, device=dev)
from pythoncrt_amd import _lib, effects
from pythoncrt_amd.pipeline import FramePipeline, baseline_config
for o in sys.argv[2:]:
    k, v = o.split("=")
    effects.DEBUG_OPTIONS[k.upper()] = int(v)
rs, h, w = baseline_config(cfg)
pipe = FramePipeline(dev, h, w, rs, noise_seed=1)
_lib.check(pipe.lib, pipe.engine.ctx, pipe.lib.crtfx_debug_buffer(pipe.engine.ctx, dbg.data_ptr()))
frames = torch.randint(0, 256, (2, h, w, 3), dtype=torch.uint8, device=dev)
pipe.run(frames)
torch.cuda.synchronize()
dbg.zero_()
pipe.run(frames)          # the planner's full launch group (4K: two frames per grid), i.e. the occupancy bench.py runs at
torch.cuda.synchronize()
d = dbg.cpu().view(-1, 8).double()
d = d[d.sum(1) > 0]
names = ["A grade->LDS", "barrier A|C2", "B H-pass", "barrier B|C1", "C1 V-pass / G", "-", "C2 masks+store", "-"]
tot = d.sum(1)
print(f"waves {len(d)}  mean cycles/wave {tot.mean():.0f}  min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(names[:7]):
    print(f"  {n:16s} {d[:, i].mean():10.0f} cyc  {100 * d[:, i].sum() / tot.sum():5.1f} %")
for wv in range(4):
    dw = d[wv::4]
    print(f"  wave {wv}: " + " ".join(f"{100 * dw[:, i].sum() / dw.sum():5.1f}" for i in range(7)))
