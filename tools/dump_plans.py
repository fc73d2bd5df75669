"""Dev tool (GPU box): what each BASELINE config (and the reference CLI's defaults) lands on — crtfx_last_plan after one small batch.
The values tests/test_plan_gpu.py pins come from here.   python tools/dump_plans.py [--opt NAME=VALUE ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pythoncrt_amd import effects  # noqa: E402
from pythoncrt_amd.pipeline import FramePipeline, baseline_config  # noqa: E402


def plans(opts=None):
    effects.DEBUG_OPTIONS = dict(opts or {})
    effects._tls.engines = {}
    dev = torch.device("cuda", 0)
    out = {}
    for cfg in (0, 2, 3, 4, 5):
        rs, h, w = baseline_config(cfg)
        half = cfg == 5
        n = 2 if half else 8
        frames = torch.zeros((n, h, w, 3), dtype=torch.float16 if half else torch.uint8, device=dev)
        pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=1, dtype=frames.dtype)
        pipe.run(frames, first_index=0)
        gm = pipe.plan().get("group_max")
        if gm:
            pipe.run(frames[:gm], first_index=0)      # exactly one planned group
        torch.cuda.synchronize()
        out["config%d" % cfg] = pipe.plan()
        del pipe, frames
        effects._tls.engines = {}
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    opts = {}
    for a in sys.argv[1:]:
        if a.startswith("--opt"):
            continue
        k, _, v = a.partition("=")
        opts[k] = int(v)
    print(json.dumps(plans(opts), indent=1))
