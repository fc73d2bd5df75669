"""Dev tool (GPU box): where does k_phosphor_ct differ from k_phosphor_cc?  Pre-warp images through an (almost) identity warp."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pythoncrt_amd as pc
from pythoncrt_amd import effects
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (150, 200)
rng = np.random.default_rng(60)
frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
tm, vg = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
out = {}
for name, opts in (("ct", {"FORCE_CC": 1}), ("cc", {"FORCE_CC": 1, "NO_CT": 1})):
    effects.DEBUG_OPTIONS.clear(); effects.DEBUG_OPTIONS.update(opts); effects._tls.engines = {}
    res = []
    for sigma in (3.0, 1.2):
        a = (frame, 0.6, tm, 2.2, False, 1, sigma, 0.25, 0.0, 1.5, vg, 2.0, 1.25, False, 1, 0, 0.0)
        res.append(pc.apply_static_effects(*a, noise_seed=7, frame_index=3, warp_strength=1e-9))
    out[name] = res
for k, (a, b) in enumerate(zip(out["ct"], out["cc"])):
    d = a != b
    print(f"sigma #{k}: {int(d.sum())} of {d.size} differ; max |d| {np.abs(a.astype(np.float64) - b).max():.3e}")
    if d.any():
        ys, xs, cs = np.nonzero(d)
        pr, pcnt = np.unique(ys, return_counts=True)
        qc, qcnt = np.unique(xs, return_counts=True)
        print("  rows:", " ".join(f"{r}:{n}" for r, n in zip(pr, pcnt)))
        print("  cols:", " ".join(f"{c}:{n}" for c, n in zip(qc, qcnt)))
        print("  channels", np.unique(cs, return_counts=True))
        for y, x, c in list(zip(ys, xs, cs))[:8]:
            print(f"   ({y},{x},{c}) ct {a[y, x, c]!r} cc {b[y, x, c]!r}")
