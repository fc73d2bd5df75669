"""What does the float32 pre-warp image cost, and what would a 16-bit one cost in accuracy?  (round 4, review item 6; a MEASUREMENT of a dev
build, -DCRTFX_PRE16: 8-byte pixels of unorm16 between k_phosphor_ct and k_warp_lean — never the product and never a parity claim.)
    python tools/pre16_error.py            # on the GPU box: runs BASELINE configs 2 and 3 at full size through the product library and
                                           # through build/ab/libcrtfx_pre16.so, compares both with the oracle and with each other
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emit(config, path):
    import torch
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    rs, h, w = baseline_config(config)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(300 + config)
    yy, xx = np.mgrid[0:h, 0:w]
    frames = []
    for i in range(2):      # smooth moving gradient + noise, as the bench's synthetic frames
        base = np.stack([(xx * 255) // (w - 1), (yy * 255) // (h - 1), ((xx + yy + 37 * i) * 255) // (h + w - 2)], axis=2)
        frames.append(np.clip((base + rng.integers(0, 256, (h, w, 3))) // 2, 0, 255).astype(np.uint8))
    frames = np.stack(frames)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=77)
    out, _ = pipe.run(torch.from_numpy(frames).to(dev), first_index=3)
    planes = []
    for i in range(2):
        t = torch.empty((h, w), dtype=torch.float32, device=dev)
        assert pipe.lib.crtfx_noise_plane(pipe.engine.ctx, 77, 3 + i, t.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        planes.append(t.cpu().numpy())
    np.savez(path, out=out.cpu().numpy(), frames=frames, planes=np.stack(planes))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--emit":
        emit(int(sys.argv[2]), sys.argv[3])
        return
    from oracle import crt_oracle as orc
    from pythoncrt_amd.pipeline import baseline_config
    tmp = os.path.join(ROOT, "gpurun_out")
    os.makedirs(tmp, exist_ok=True)
    res = {}
    for config in (2, 3):
        paths = {}
        for name, lib in (("product", None), ("pre16", os.path.join(ROOT, "build", "ab", "libcrtfx_pre16.so"))):
            env = dict(os.environ)
            if lib:
                env["CRTFX_LIB"] = lib
            paths[name] = os.path.join(tmp, f"pre16_{name}_c{config}.npz")
            subprocess.run([sys.executable, os.path.abspath(__file__), "--emit", str(config), paths[name]], env=env, check=True)
        a, b = np.load(paths["product"]), np.load(paths["pre16"])
        rs, h, w = baseline_config(config)
        params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength",
                                              "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "warp_strength")}
        exp, _ = orc.process_frames(list(a["frames"]), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                    rs.vignette_strength, noise_planes=list(a["planes"]), first_index=3)
        exp = np.stack(exp).astype(np.int16)

        def cmp(x):
            d = np.abs(x.astype(np.int16) - exp)
            return {"max_lsb": int(d.max()), "frac_off": float((d != 0).mean())}
        dd = np.abs(a["out"].astype(np.int16) - b["out"].astype(np.int16))
        res[f"config{config}"] = {"size": [h, w], "product_vs_oracle": cmp(a["out"]), "pre16_vs_oracle": cmp(b["out"]),
                                  "pre16_vs_product": {"max_lsb": int(dd.max()), "frac_off": float((dd != 0).mean())}}
        for p_ in paths.values():
            os.unlink(p_)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
