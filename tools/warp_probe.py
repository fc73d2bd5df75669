"""Dev tool (GPU box): k_phosphor / k_warp time per frame for uint8 and half frames at 4K and 8K (BASELINE config 3 / 5 settings) — what of the 8K half
warp's 138 us is the frame size (pre-warp image past the Infinity Cache) and what the 6-byte output pixels?     python tools/warp_probe.py [--opt K=V ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pythoncrt_amd import effects
from pythoncrt_amd.pipeline import FramePipeline, baseline_config

effects.DEBUG_OPTIONS = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
dev = torch.device("cuda", 0)
rs = baseline_config(3)[0]
for (h, w, n) in ((2160, 3840, 32), (4320, 7680, 8)):
    for dtype in (torch.uint8, torch.float16):
        frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev).to(dtype)
        pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=1, dtype=dtype)
        out = torch.empty_like(frames)
        for _ in range(3):
            pipe.run(frames, out=out)
        torch.cuda.synchronize()
        pipe.profile(1)
        for i in range(4):
            pipe.run(frames, first_index=i * n, out=out)
        torch.cuda.synchronize()
        k = pipe.profile_read()
        pipe.profile(False)
        per = {name: v[0] * v[1] / v[2] * 1e3 for name, v in k.items() if v[1] and v[2]}
        print(f"{w}x{h} {'half ' if dtype == torch.float16 else 'uint8'}: " + "  ".join(f"{name} {us:7.1f} us/frame" for name, us in per.items()) +
              f"   ({sum(per.values()) / (h * w / 8294400):6.1f} us per 4K-equivalent)  {pipe.plan().get('phosphor')} | {pipe.plan().get('warp')}", flush=True)
        del pipe, frames, out
        torch.cuda.empty_cache()
