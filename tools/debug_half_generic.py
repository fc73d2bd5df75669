"""Dev tool (GPU box): half frames, FORCE_GENERIC against the default builds through the render loop — which stage makes them differ?"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pythoncrt_amd import effects
from pythoncrt_amd.pipeline import FramePipeline, baseline_config
dev = torch.device("cuda", 0)
h, w = 40, 448
rng = np.random.default_rng(901)
f16 = torch.from_numpy((rng.random((3, h, w, 3), dtype=np.float32) * 255.0).astype(np.float16)).to(dev)
u8 = torch.from_numpy(rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)).to(dev)
base = dataclasses.replace(baseline_config(5)[0], warp_strength=0.0)
cases = {"full": {}, "no_noise": dict(noise_strength=0.0), "no_vig": dict(vignette_strength=0.0), "no_triad": dict(triad_strength=0.0),
         "no_scan": dict(scanline_strength=0.0), "noise_only": dict(vignette_strength=0.0, triad_strength=0.0, scanline_strength=0.0, aberration_px=0),
         "noise_nobloom": dict(vignette_strength=0.0, triad_strength=0.0, scanline_strength=0.0, aberration_px=0, bloom_strength=0.0)}
for fname, frames in (("f16", f16), ("u8", u8)):
    for cname, kw in cases.items():
        rs = dataclasses.replace(base, **kw)
        outs = {}
        for name, opts in (("def", {}), ("generic", {"FORCE_GENERIC": 1}), ("runtime", {"FORCE_RUNTIME_FLAGS": 1})):
            effects.DEBUG_OPTIONS = dict(opts); effects._tls.engines = {}
            pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=99, dtype=frames.dtype)
            out, _ = pipe.run(frames, first_index=4)
            outs[name] = out.float().cpu().numpy()
            pl = pipe.plan()
        for name in ("generic", "runtime"):
            d = outs["def"] != outs[name]
            idx = np.argwhere(d)
            print(fname, cname, name, "ndiff", len(idx), idx[:3].tolist(), [(float(outs["def"][tuple(i)]), float(outs[name][tuple(i)])) for i in idx[:3]], pl.get("phosphor", pl.get("point")))
