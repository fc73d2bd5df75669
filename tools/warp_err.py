import numpy as np, sys
sys.path.insert(0, '/root/repo')
import pythoncrt_amd as pc
from oracle import crt_oracle as orc
h, w = 270, 480
rng = np.random.default_rng(5)
worst = 0; mism = []
for seed in range(4):
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    plane = rng.standard_normal((h, w), dtype=np.float32)
    a = lambda tm, vg: (frame, 0.6, tm, 2.2, False, 1, 3.0, 0.25, 0.0, 1.5, vg, 2.0, 1.25, False, 1, 0, 0.0)
    g = pc.apply_static_effects(*a(pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)), warp_strength=0.15, noise_plane=plane)
    o = orc.apply_static_effects(*a(orc.make_triad_mask(h, w, 0.35, 0.5), orc.make_vignette(h, w, 0.25)), warp_strength=0.15, noise_plane=plane)
    worst = max(worst, float(np.abs(g.astype(np.float64) - o).max()))
    d = orc.convert_scale_abs(g) != orc.convert_scale_abs(o)
    mism.append(float(d.mean()))
print("max float err %.3e  u8 mismatch rate %.2e" % (worst, np.mean(mism)))
