"""Which kernel build each BASELINE config lands on (crtfx_last_plan).

Every variant of a kernel produces the same bits (test_kernel_variants_agree, test_fp16_column_owner_kernel), so a planner edit that silently
moves a config from its tuned build to a general one stays green in every parity test and only shows as a slower bench line a round later.
This file pins the plan: the expected strings are what `tools/dump_plans.py` printed for the round-5 sources (gpurun_out/r05_plans.json);
change them together with a deliberate planner change, never to make the test pass."""
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

WARP_PLAIN_U8 = "k_warp_lean<f64,none,u8,rows=4,tile=128x8,plain>"
SEG_1080P = 168
EXPECTED = {
    # config: (frames to run, expected key -> value)
    2: (8, dict(phosphor="k_phosphor_ct<4,u8>", group=5, seg_rows=SEG_1080P, warp=WARP_PLAIN_U8, warp_frames=5)),
    3: (4, dict(phosphor="k_phosphor_ct<9,u8>", group=2, seg_rows=256, warp=WARP_PLAIN_U8, warp_frames=2)),
    4: (8, dict(phosphor="k_phosphor_ct<4,u8>", group=5, seg_rows=SEG_1080P, warp="k_warp_lean<f64,render,u8,rows=2,tile=64x8,plain>", warp_frames=4)),      # the clip's first frame passes through unblended (ref:1094-1095): a run of 4
    5: (2, dict(phosphor="k_phosphor_ct<9,half>", group=1, seg_rows=256, warp="k_warp_lean<f64,none,half,rows=4,tile=128x8,plain>", warp_frames=1)),
    0: (8, dict(point="k_point_fused_seq<fast+pixelate,u8,render>")),      # the reference CLI's defaults (round 6: the half-resolution bloom source is formed inside the pointwise kernel)
}


def _plan(cfg, opts, monkeypatch):
    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    dev = torch.device("cuda", torch.cuda.current_device())
    monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(opts))
    effects._tls.engines = {}
    rs, h, w = baseline_config(cfg)
    n = EXPECTED[cfg][0]
    frames = torch.zeros((n, h, w, 3), dtype=torch.float16 if cfg == 5 else torch.uint8, device=dev)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=1, dtype=frames.dtype)
    pipe.run(frames, first_index=0)
    gm = pipe.plan().get("group_max")
    if gm:                                              # a batch of exactly one planned group: its launch is the one the plan describes
        pipe.run(frames[:gm], first_index=0)
    torch.cuda.synchronize()
    plan = pipe.plan()
    del pipe, frames
    effects._tls.engines = {}
    torch.cuda.empty_cache()
    return plan


@pytest.mark.parametrize("cfg", [2, 3, 4, 5, 0])
def test_baseline_configs_land_on_their_builds(cfg, monkeypatch):
    plan = _plan(cfg, {}, monkeypatch)
    for key, want in EXPECTED[cfg][1].items():
        assert plan.get(key) == want, (cfg, key, plan)
    if cfg == 0:
        assert plan.get("group", 0) >= 2 and "phosphor" not in plan and "warp" not in plan and "half" not in plan, plan


@pytest.mark.parametrize("cfg,opts,key", [(3, {"NO_CT": 1}, "phosphor"), (5, {"NO_CT": 1}, "phosphor"), (2, {"NO_CT": 1}, "phosphor"),
                                          (3, {"NO_PLAIN_WARP": 1}, "warp"), (5, {"NO_PLAIN_WARP": 1}, "warp"), (4, {"NO_PLAIN_WARP": 1}, "warp"),
                                          (3, {"FORCE_GENERIC": 1}, "phosphor"), (0, {"FORCE_RUNTIME_FLAGS": 1}, "point"), (0, {"NO_FUSED_HALF": 1}, "point")])
def test_a_forced_fallback_is_seen(cfg, opts, key, monkeypatch):
    """The guard guards: with a build switched off the recorded plan differs from the pinned one (so the test above would fail)."""
    plan = _plan(cfg, opts, monkeypatch)
    assert plan.get(key) and plan.get(key) != EXPECTED[cfg][1][key], (cfg, opts, plan)


def test_last_plan_through_the_c_abi(monkeypatch):
    """Buffer handling of crtfx_last_plan: truncation to n - 1 characters, NUL termination, argument checks."""
    import ctypes
    from pythoncrt_amd import _lib, effects
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    dev = torch.device("cuda", torch.cuda.current_device())
    rs, _, _ = baseline_config(3)
    h, w = 64, 128
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=1)
    lib, ctx = pipe.lib, pipe.engine.ctx
    buf = ctypes.create_string_buffer(512)
    assert lib.crtfx_last_plan(ctx, buf, len(buf)) == _lib.OK and buf.value == b""          # nothing launched yet
    pipe.run(torch.zeros((2, h, w, 3), dtype=torch.uint8, device=dev))
    assert lib.crtfx_last_plan(ctx, buf, len(buf)) == _lib.OK
    full = buf.value
    assert full.startswith(b"phosphor=k_phosphor_ct<9,u8>;") and b";warp=k_warp_lean<" in full
    small = ctypes.create_string_buffer(b"\xff" * 16, 16)
    assert lib.crtfx_last_plan(ctx, small, 12) == _lib.OK
    assert small.value == full[:11] and small.raw[12:] == b"\xff" * 4                       # n - 1 characters + NUL, nothing past n
    assert lib.crtfx_last_plan(ctx, None, 12) == _lib.E_INVALID and lib.crtfx_last_plan(ctx, small, 0) == _lib.E_INVALID
    assert lib.crtfx_last_plan(None, small, 12) == _lib.E_INVALID


def test_plan_of_a_batch_with_a_tail_is_the_full_group(monkeypatch):
    """crtfx_last_plan after a batch that is not a multiple of the group size (round 6): the record is the FULL-SIZE launch group's, not the
    shorter tail group's — bench.py's 8192-frame 1080p step is 1638 groups of 5 frames x 168-row blocks + one of 2 x 64, and round 5's
    committed lines read `group 2, seg_rows 64`."""
    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    dev = torch.device("cuda", torch.cuda.current_device())
    monkeypatch.setattr(effects, "DEBUG_OPTIONS", {})
    effects._tls.engines = {}
    rs, h, w = baseline_config(2)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=1)
    frames = torch.zeros((7, h, w, 3), dtype=torch.uint8, device=dev)      # 5 + 2
    pipe.run(frames, first_index=0)
    plan = pipe.plan()
    assert plan["group"] == 5 and plan["seg_rows"] == SEG_1080P and plan["warp_frames"] == 5 and plan["group_max"] == 5, plan
    pipe.run(frames[:2], first_index=0)                                    # a batch that IS only the short group reports that one
    tail = pipe.plan()
    assert tail["group"] == 2 and tail["seg_rows"] != SEG_1080P and tail["warp_frames"] == 2, tail
    del pipe, frames
    effects._tls.engines = {}
    torch.cuda.empty_cache()
