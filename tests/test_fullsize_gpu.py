"""Full-size oracle parity for the BASELINE configs that round 2 only met at reduced sizes, and the render loop with
its per-frame records built ahead of the launches.

  * configs[3]: 1080p, full chain, persistence 0.5 — eight frames through crtfx_process_batch, so that the planner's
    multi-frame k_phosphor group and the in-register persistence chain of k_warp_lean run at the size they are planned
    for, against the oracle's in-order render (crt_filter.py ref:1086-1098).
  * configs[4]: one 8K frame held as float16 — the 688-row / multi-round launch shape of the register-window kernel's
    half build, block seams and all — against the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import crt_oracle as orc  # noqa: E402  (checker only)
from tests.test_parity_gpu import _export_planes, make_frame  # noqa: E402

PARAM_KEYS = ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength",
              "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "warp_strength")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda", torch.cuda.current_device())


def test_1080p_persistence_render_loop_against_oracle(dev):
    """BASELINE configs[3] at full size on one GPU: 8 frames = one full planner group + a partial one; frame 0 passes
    through unblended (ref:1094-1095), frames 1.. blend in order.  <= 1 LSB, < 0.1 % of the samples, as the 4K test."""
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    rs, h, w = baseline_config(4)
    assert (h, w) == (1080, 1920) and rs.persistence == 0.5
    n, first, seed = 8, 11, 31337
    frames = np.stack([make_frame(h, w, seed=500 + i, kind="grad" if i % 2 else "noise") for i in range(n)])
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=seed)
    out, state = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    planes = _export_planes(pipe, seed, first, n, h, w)
    params = {k: getattr(rs, k) for k in PARAM_KEYS}
    exp, exp_state = orc.process_frames(list(frames), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength,
                                        rs.triad_softness, rs.vignette_strength, noise_planes=planes, first_index=first)
    got = out.cpu().numpy()
    for i in range(n):
        d = np.abs(got[i].astype(np.int16) - exp[i].astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, (i, int(d.max()), float((d != 0).mean()))
    assert np.abs(state.cpu().numpy().astype(np.float64) - exp_state).max() <= 1e-6
    # the same clip in two calls (a chunk boundary carries the state through HBM instead of registers): identical frames
    a, st = pipe.run(torch.from_numpy(frames[:3]).to(dev), first_index=first)
    b, _ = pipe.run(torch.from_numpy(frames[3:]).to(dev), first_index=first + 3, state=st)
    assert torch.equal(torch.cat([a, b]), out)


def test_8k_fp16_frame_against_oracle(dev):
    """BASELINE configs[4]: one 7680 x 4320 frame of float16 pixels through the render loop against the oracle (about a
    minute of CPU).  The half output is |x * 255| narrowed (convertScaleAbs without the integer rounding)."""
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    rs, h, w = baseline_config(5)
    assert (h, w) == (4320, 7680)
    first, seed = 2, 808
    rng = np.random.default_rng(77)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    grad = np.stack([xx * (255.0 / w), yy * (255.0 / h), (xx + yy) * (255.0 / (h + w))], axis=2)
    frame = ((grad + rng.random((h, w, 3), dtype=np.float32) * 255.0) * 0.5).astype(np.float16)      # fractional values on the 0..255 scale
    del grad, yy, xx
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=seed, dtype=torch.float16)
    out, _ = pipe.run(torch.from_numpy(frame[None]).to(dev), first_index=first)
    got = out[0].cpu().numpy()
    planes = _export_planes(pipe, seed, first, 1, h, w)
    params = {k: getattr(rs, k) for k in PARAM_KEYS}
    _, st = orc.process_frames([frame], params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                               rs.vignette_strength, noise_planes=planes, first_index=first)
    exp16 = np.abs(st.astype(np.float32) * np.float32(255.0)).astype(np.float16)
    assert got.dtype == np.float16 and got.shape == exp16.shape
    diff = np.abs(got.astype(np.float32) - exp16.astype(np.float32))
    assert diff.max() <= 0.125, float(diff.max())                    # one half ulp at 128..255
    assert (got != exp16).mean() < 5e-3, float((got != exp16).mean())     # half is 32x finer than uint8 around 200


@pytest.mark.parametrize("ab,hw,sigma", [(1, (40, 448), 3.0), (0, (33, 702), 1.2), (-3, (40, 448), 3.0), (8, (24, 640), 4.4), (1, (30, 449), 3.0), (-8, (40, 320), 3.0)])
def test_fp16_wide_frames_against_oracle(dev, ab, hw, sigma):
    """Half frames wide enough to have interior strips (the small fp16 cases of test_parity_gpu.py are all edge strips), every aberration
    sign and size, three radii, an odd width: the half output frame of the gate-folded k_phosphor_rr<half> against the oracle's float image
    narrowed the same way, and the same frames on an allocation that does not start on a dword.  (A pixel-pair A phase for this build —
    one 12-byte load per two pixels instead of six 2-byte loads — was measured 17 % SLOWER in round 4, profiles/r04_8k_bands.txt, and is
    not in the tree; this test is what held it to the oracle.)"""
    import dataclasses
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    h, w = hw
    rs = dataclasses.replace(baseline_config(5)[0], aberration_px=ab, bloom_sigma=sigma, warp_strength=0.0)
    first, seed, n = 4, 99, 2
    rng = np.random.default_rng(500 + ab)
    frames = (rng.random((n, h, w, 3), dtype=np.float32) * 255.0).astype(np.float16)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=seed, dtype=torch.float16)
    out, _ = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    got = out.cpu().numpy()
    planes = _export_planes(pipe, seed, first, n, h, w)
    params = {k: getattr(rs, k) for k in PARAM_KEYS}
    for i in range(n):
        _, st = orc.process_frames([frames[i]], params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                   rs.vignette_strength, noise_planes=[planes[i]], first_index=first + i)
        exp16 = np.abs(st.astype(np.float32) * np.float32(255.0)).astype(np.float16)
        diff = np.abs(got[i].astype(np.float32) - exp16.astype(np.float32))
        assert diff.max() <= 0.125 and (got[i] != exp16).mean() < 5e-3, (ab, hw, i, float(diff.max()), float((got[i] != exp16).mean()))      # the bar of the 8K test
    # the same frames two bytes into an allocation (not on a dword): the per-sample path, the same bits
    raw = torch.zeros(frames.size * 2 + 8, dtype=torch.uint8, device=dev)
    shifted = raw[2:2 + frames.size * 2].view(torch.float16).view(n, h, w, 3)
    shifted.copy_(torch.from_numpy(frames).to(dev))
    assert shifted.data_ptr() % 4 == 2
    out2, _ = pipe.run(shifted, first_index=first)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("ab,hw,sigma", [(1, (40, 448), 3.0), (0, (33, 704), 1.2), (-3, (40, 448), 3.0), (8, (24, 640), 4.0), (-8, (40, 320), 3.0), (2, (60, 256), 0.4),
                                         (1, (30, 449), 3.0), (1, (40, 450), 3.0), (-1, (37, 708), 2.0), (5, (1100, 128), 3.3), (1, (520, 1024), 3.0),
                                         (1, (40, 448), 4.4), (-2, (48, 512), 4.6), (3, (64, 576), 5.0),      # radii 13 - 15: the half build with spilled registers
                                         (1, (36, 100), 3.0), (-2, (28, 212), 1.2)])      # last strips of 36 / 20 pixels: partial qword rows out of k_warp_lean
def test_fp16_column_owner_kernel(dev, ab, hw, sigma, monkeypatch):
    """Round 5: half frames that park a pre-warp image (warp on) run k_phosphor_ct<R, half> — frame-row windows as aligned qwords, the raw units of a
    trip in a one-trip LDS tile, each consumer thread's centre samples in a register window of packed halves.  Every aberration sign and size,
    radii 1 .. 15, interior / edge / partial strips, a width that is not a multiple of four (all strips on the per-sample A phase), an odd
    width (frames of a batch off a qword: the register-window kernel), tall frames (several row segments, window-fill trips), and a frame that does not start on a qword (falls back to k_phosphor_rr):
    (1) bit-identical frames to the register-window kernel (NO_CT), its runtime-gate instantiation and the generic LDS-ring kernel;
    (2) the oracle at the bar of the 8K test."""
    import dataclasses
    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    h, w = hw
    rs = dataclasses.replace(baseline_config(5)[0], aberration_px=ab, bloom_sigma=sigma)
    assert rs.warp_strength == 0.15
    first, seed, n = 4, 99, 3
    rng = np.random.default_rng(900 + ab)
    frames = (rng.random((n, h, w, 3), dtype=np.float32) * 255.0).astype(np.float16)
    frames[0, :, : w // 2] = frames[0, :1, :1]                       # flat areas beside noise
    frames[1, h // 3] = np.float16(255.0)
    frames[2, :, ::7] = np.float16(0.0)
    dframes = torch.from_numpy(frames).to(dev)
    outs = {}
    for name, opts in (("ct", {}), ("rr", {"NO_CT": 1}), ("runtime", {"FORCE_RUNTIME_FLAGS": 1}), ("generic", {"FORCE_GENERIC": 1}), ("ct_seg", {"SEG_ROWS": 24, "GROUP": 2})):
        monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(opts))
        effects._tls.engines = {}
        pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=seed, dtype=torch.float16)
        out, _ = pipe.run(dframes, first_index=first)
        outs[name] = out.clone()
        if name == "ct":
            # (a frame size that is not a multiple of 8 bytes — 30 x 449 — puts the batch's second frame off a qword: the whole group falls back)
            want = "k_phosphor_ct<%d,half>" % rs_radius(sigma) if (h * w * 6) % 8 == 0 else "k_phosphor_rr<"
            assert pipe.plan().get("phosphor", "").startswith(want), pipe.plan()
            planes = _export_planes(pipe, seed, first, n, h, w)
            # two bytes into an allocation: not on a qword -> the register-window kernel, the same bits
            raw = torch.zeros(frames.size * 2 + 8, dtype=torch.uint8, device=dev)
            shifted = raw[2:2 + frames.size * 2].view(torch.float16).view(n, h, w, 3)
            shifted.copy_(dframes)
            out2, _ = pipe.run(shifted, first_index=first)
            assert pipe.plan().get("phosphor", "").startswith("k_phosphor_rr<"), pipe.plan()
            assert torch.equal(out2, out)
        elif name == "rr":
            assert pipe.plan().get("phosphor", "").startswith("k_phosphor_rr<"), pipe.plan()
    effects._tls.engines = {}
    for name in ("rr", "runtime", "generic", "ct_seg"):
        assert torch.equal(outs["ct"], outs[name]), name
    got = outs["ct"].cpu().numpy()
    params = {k: getattr(rs, k) for k in PARAM_KEYS}
    for i in range(n if h * w < 200000 else 1):
        _, st = orc.process_frames([frames[i]], params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                   rs.vignette_strength, noise_planes=[planes[i]], first_index=first + i)
        exp16 = np.abs(st.astype(np.float32) * np.float32(255.0)).astype(np.float16)
        diff = np.abs(got[i].astype(np.float32) - exp16.astype(np.float32))
        assert diff.max() <= 0.125 and (got[i] != exp16).mean() < 5e-3, (ab, hw, i, float(diff.max()), float((got[i] != exp16).mean()))


def rs_radius(sigma):
    return int(round(3 * sigma))


@pytest.mark.parametrize("speed", [30.0, 31.7])       # integer phases (one shared table, regenerated far ahead) / fractional phases (one table per batch)
def test_records_built_ahead_of_their_launches(dev, speed):
    """frame_records() for several batches BEFORE the first of them runs (bench.py --tables-outside, GpuShardEngine.records):
    every batch's scanline row-gain table must stay alive with its records — also across a forced regeneration of the
    shared integer-phase table — and give the frames of the build-then-run order."""
    import dataclasses
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    rs = dataclasses.replace(baseline_config(2)[0], scanline_speed_px_s=speed, warp_strength=0.0)
    h, w, n = 72, 136, 5
    firsts = [0, 5, 200000, 400000, 10]          # 200000 / 400000: beyond the shared table's look-ahead -> regenerated twice
    frames = torch.from_numpy(np.stack([make_frame(h, w, seed=700 + i, kind="grad") for i in range(n)])).to(dev)
    ref_pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=5)
    want = [ref_pipe.run(frames, first_index=f)[0].clone() for f in firsts]
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=5)
    ahead = [pipe.frame_records(f, n) for f in firsts]
    # churn the allocator with same-sized blocks between building and running: a freed table would be handed out again
    junk = [torch.full((n, h), float(k), dtype=torch.float32, device=dev) for k in range(64)]
    big = [torch.zeros(200000 + h + 65536 + 16, dtype=torch.float32, device=dev) for _ in range(4)]
    for f, recs, exp in zip(firsts, ahead, want):
        got, _ = pipe.run(frames, first_index=f, records=recs)
        assert torch.equal(got, exp), f
    del junk, big


CLI_PARAM_KEYS = PARAM_KEYS + ("brightness", "contrast", "gamma", "saturation", "temperature", "flicker_strength", "flicker_hz", "grain_size")


@pytest.mark.parametrize("name,kw,build", [
    ("defaults", {}, "k_point_fused_seq<fast+pixelate,u8,render>"),
    ("grade table", dict(brightness=0.05, contrast=1.1, gamma=1.2, temperature=-0.2), "k_point_fused_seq<fast+pixelate+gradelut,u8,render>"),
    ("saturation", dict(saturation=1.3), "k_point_fused_seq<fast+pixelate+sat,u8,render>"),
    ("coarse grain", dict(grain_size=2), "k_point_fused_seq<fast+pixelate+coarse,u8,render>"),
    ("several knobs", dict(saturation=1.2, bloom_threshold=0.3, flicker_strength=0.2, flicker_hz=9.0, pixel_size=1), "k_point_fused_seq<fast+grade,u8,render>"),
    ("stages off", dict(vignette_strength=0.0, noise_strength=0.0, persistence=0.0), "k_point_fused_seq<runtime,u8,none>"),
])
def test_1080p_cli_chain_on_the_fused_kernel_against_oracle(dev, name, kw, build):
    """The chain `python -m pythoncrt_amd.cli` runs with no flags (ref:1160-1206: fast bloom, pixel size 2, persistence 0.2) and with one or several
    knobs turned, at 1080p through crtfx_process_batch — round 6's k_point_fused_seq in each of its gate forms, its half-resolution tiles, prologue
    and a run of eight frames at the size bench.py --config 0 measures — against the oracle's in-order render of the same ten frames, the grain the
    kernels drew exported for it.  Frame 0 (no blend yet) bit-exact; the blended frames <= 1 LSB on < 0.1 % of the samples."""
    import dataclasses
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    rs = dataclasses.replace(RenderSettings(), **kw)
    h, w, n, first, seed = 1080, 1920, 10, 5, 777
    frames = np.stack([make_frame(h, w, seed=900 + i, kind="grad" if i % 2 else "noise") for i in range(n)])
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=seed)
    out, state = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    assert pipe.plan().get("point") == build, pipe.plan()
    planes = None
    if rs.noise_strength > 0.0:
        g = rs.grain_size
        if g > 1:
            from pythoncrt_amd.effects import Engine
            small = Engine(dev, max(1, h // g), max(1, w // g), 0)
            planes = []
            for i in range(n):
                p = torch.empty((small.h, small.w), dtype=torch.float32, device=dev)
                assert small.lib.crtfx_noise_plane(small.ctx, seed, first + i, p.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
                planes.append(p.cpu().numpy())
        else:
            planes = _export_planes(pipe, seed, first, n, h, w)
    params = {k: getattr(rs, k) for k in CLI_PARAM_KEYS}
    exp, exp_state = orc.process_frames(list(frames), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength,
                                        rs.triad_softness, rs.vignette_strength, noise_planes=planes, first_index=first)
    got = out.cpu().numpy()
    assert np.array_equal(got[0], exp[0]), name
    for i in range(n):
        d = np.abs(got[i].astype(np.int16) - exp[i].astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, (name, i, int(d.max()), float((d != 0).mean()))
    if state is not None:
        assert np.abs(state.cpu().numpy().astype(np.float64) - exp_state).max() <= 1e-6, name
