"""GPU: a seeded random sweep of the render loop (FramePipeline = crtfx_process_batch, in-kernel grain RNG, so the
gate-folded / lean kernels are the ones that run) against the oracle's in-order render, over ragged frame sizes
(narrower than a 64-px strip, shorter than an 8-row block, odd widths) and the whole settings space of the CLI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import crt_oracle as orc  # noqa: E402  (checker only)


def draw_settings(rng):
    from pythoncrt_amd.pipeline import RenderSettings
    pick = lambda *a: a[int(rng.integers(len(a)))]
    fast = bool(rng.integers(2))
    return RenderSettings(
        scanline_strength=pick(0.0, 0.6, 1.0), triad_strength=pick(0.0, 0.35, 0.9), triad_gamma=pick(2.2, 1.0, 0.7),
        triad_preserve_luma=bool(rng.integers(2)), triad_softness=pick(0.0, 0.5, 1.4), aberration_px=int(pick(-8, -1, 0, 1, 3)),
        bloom_sigma=pick(0.0, 0.5, 1.2, 3.0, 4.4, 11.0, 25.0), bloom_strength=pick(0.0, 0.25, 0.8), bloom_threshold=pick(0.0, 0.0, 0.3),
        noise_strength=pick(0.0, 1.5, 6.0), vignette_strength=pick(0.0, 0.25, 1.0), persistence=pick(0.0, 0.2, 0.9),
        scanline_speed_px_s=pick(30.0, 0.0, -12.5), scanline_period_px=pick(2.0, 3.7), fast_bloom=fast, pixel_size=int(pick(1, 1, 2, 3)),
        brightness=pick(0.0, 0.0, 0.08), contrast=pick(1.0, 1.0, 1.25), gamma=1.0, saturation=pick(1.0, 1.0, 1.4), temperature=pick(0.0, 0.0, -0.5),
        flicker_strength=pick(0.0, 0.0, 0.5), flicker_hz=pick(0.0, 9.0), grain_size=int(pick(1, 1, 1, 2)),
        scanline_angle=pick(0.0, 0.0, 0.0, 12.0), scanline_thickness=pick(1.0, 1.0, 1.0, 2.0), warp_strength=pick(0.0, 0.15, 0.15, -0.3, 0.6),
        glitch_amp_px=int(pick(0, 0, 0, 7)), glitch_height_frac=pick(0.0, 0.3))


SIZES = [(1, 1), (2, 3), (7, 65), (9, 200), (33, 63), (64, 64), (37, 129), (90, 160), (17, 300), (130, 70)]


@pytest.mark.parametrize("case", range(150))
def test_random_render_matches_oracle(case):
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from pythoncrt_amd.pipeline import FramePipeline
    rng = np.random.default_rng(1000 + case)
    h, w = SIZES[case % len(SIZES)]
    rs = draw_settings(rng)
    n, first, fps, seed = 6, int(rng.integers(0, 50)), 25.0, int(rng.integers(1 << 40))      # 6 frames: a full 4-frame persistence run behind frame 0, then a partial one
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    dev = torch.device("cuda", torch.cuda.current_device())
    ov_mode = int(rng.integers(4))              # 0, 1: no text overlay; 2: blended before the effects; 3: after them (ref:588-598 / 653-663)
    ov = None
    if ov_mode >= 2:
        ov = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        ov[..., 3] = np.where(rng.random((h, w)) < 0.5, 0, ov[..., 3])      # half the plane untouched, the rest any alpha
    pipe = FramePipeline(dev, h, w, rs, fps=fps, noise_seed=seed, text_overlay_rgba=ov, text_overlay_after=(ov_mode == 3))
    out, state = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    # the grain the kernels drew, exported for the oracle (cv2.randn is unreproducible)
    gh, gw = (h, w) if rs.grain_size <= 1 else (max(1, h // rs.grain_size), max(1, w // rs.grain_size))
    planes = None
    if rs.noise_strength > 0.0:
        planes = []
        from pythoncrt_amd.effects import Engine
        small = Engine(dev, gh, gw, 0) if (gh, gw) != (h, w) else pipe.engine
        for i in range(n):
            p = torch.empty((gh, gw), dtype=torch.float32, device=dev)
            rc = small.lib.crtfx_noise_plane(small.ctx, seed, first + i, p.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            planes.append(p.cpu().numpy())
    params = {k: getattr(rs, k) for k in (
        "scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength", "bloom_threshold",
        "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "glitch_amp_px", "glitch_height_frac", "brightness", "contrast",
        "gamma", "saturation", "temperature", "flicker_strength", "flicker_hz", "grain_size", "scanline_angle", "scanline_thickness",
        "warp_strength")}
    if ov is not None:
        params["text_overlay_rgba"], params["text_overlay_after"] = ov, ov_mode == 3
    exp, exp_state = orc.process_frames(list(frames), params, fps, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength,
                                        rs.triad_softness, rs.vignette_strength, noise_planes=planes, first_index=first)
    got = out.cpu().numpy()
    d = np.abs(got.astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1, (case, rs, int(d.max()))
    assert (d != 0).mean() <= max(2e-3, 2.0 / d.size), (case, rs, float((d != 0).mean()))
    if rs.persistence > 0.0:
        assert state is not None and np.abs(state.cpu().numpy().astype(np.float64) - exp_state).max() <= 1e-6


@pytest.mark.parametrize("hw", [(6, 20011), (20011, 6), (3, 32767)])
def test_extreme_aspect_frames(hw):
    """Very wide / very tall frames (the ctx accepts up to 32767 per side): strip and segment arithmetic, 32-bit
    offsets and the warp's short-saturated tap origins against the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    h, w = hw
    rng = np.random.default_rng(h * 7 + w)
    rs = RenderSettings(fast_bloom=False, bloom_sigma=1.2, pixel_size=1, persistence=0.3, warp_strength=0.15, noise_strength=0.0)
    frames = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    dev = torch.device("cuda", torch.cuda.current_device())
    pipe = FramePipeline(dev, h, w, rs, fps=25.0, noise_seed=1)
    out, state = pipe.run(torch.from_numpy(frames).to(dev))
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma",
                                          "bloom_strength", "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom",
                                          "pixel_size", "warp_strength")}
    exp, _ = orc.process_frames(list(frames), params, 25.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                rs.vignette_strength)
    d = np.abs(out.cpu().numpy().astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (hw, int(d.max()), float((d != 0).mean()))


CT_SIZES = [(40, 704), (26, 1028), (9, 512), (64, 322), (17, 960), (130, 260), (8, 4096), (33, 641)]


@pytest.mark.parametrize("case", range(32))
def test_random_full_chain_on_the_headline_kernel(case):
    """The gate set of BASELINE configs 2-5 (scanlines + triad LUTs + Gaussian bloom + vignette + grain, warp or persistence behind
    it) is what k_phosphor_ct serves: 32 seeded draws of everything that varies INSIDE that gate set — bloom radius 1..15, strengths,
    triad strength / softness (two-valued, three-valued and unsoftened masks), aberration -8..8, scanline period / phase, frame sizes with
    interior strips, edge strips, partial last strips and widths that are not a multiple of four — through the render loop against the
    oracle's in-order render.  No warp: frame 0 bit-exact (it is the quantised pre-warp image) and the blended frames <= 1 LSB."""
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    rng = np.random.default_rng(7000 + case)
    pick = lambda *a: a[int(rng.integers(len(a)))]
    h, w = CT_SIZES[case % len(CT_SIZES)]
    warp = pick(0.0, 0.0, 0.15)
    rs = RenderSettings(
        scanline_strength=pick(0.3, 0.6, 1.0), triad_strength=pick(0.2, 0.35, 0.5, 1.0), triad_gamma=pick(2.2, 1.8, 0.6), triad_preserve_luma=False,
        triad_softness=pick(0.0, 0.5, 1.0, 2.0), aberration_px=int(rng.integers(-8, 9)), bloom_sigma=pick(0.2, 0.5, 1.0, 1.2, 2.0, 2.7, 3.0, 3.7, 4.0, 4.4, 4.6, 5.0),      # radii 1 .. 15: all on k_phosphor_ct (13 .. 15 with spilled registers)
        bloom_strength=pick(0.1, 0.25, 0.9), bloom_threshold=0.0, noise_strength=pick(0.5, 1.5, 6.0), vignette_strength=pick(0.1, 0.25, 1.0),
        persistence=(pick(0.2, 0.5) if warp == 0.0 else pick(0.0, 0.5)), scanline_speed_px_s=pick(30.0, -12.5, 7.0), scanline_period_px=pick(2.0, 3.7),
        fast_bloom=False, pixel_size=1, warp_strength=warp)
    n, first, fps, seed = 4, int(rng.integers(0, 40)), 25.0, int(rng.integers(1 << 40))
    kind = rng.integers(3)
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    if kind == 1:
        frames[:] = (np.arange(w, dtype=np.int64)[None, None, :, None] * 7 + np.arange(h)[None, :, None, None] * 3 + np.arange(3)[None, None, None, :] * 50) % 256
    elif kind == 2:
        frames[:, :, : w // 2] = 255          # saturated half: bloom clips, LUT index 1024
    dev = torch.device("cuda", torch.cuda.current_device())
    pipe = FramePipeline(dev, h, w, rs, fps=fps, noise_seed=seed)
    out, state = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    planes = []
    for i in range(n):
        p = torch.empty((h, w), dtype=torch.float32, device=dev)
        assert pipe.lib.crtfx_noise_plane(pipe.engine.ctx, seed, first + i, p.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        planes.append(p.cpu().numpy())
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength",
                                          "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "warp_strength")}
    exp, exp_state = orc.process_frames(list(frames), params, fps, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                        rs.vignette_strength, noise_planes=planes, first_index=first)
    got = out.cpu().numpy()
    if warp == 0.0:
        assert np.array_equal(got[0], exp[0]), (case, rs)
    d = np.abs(got.astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() <= 2e-3, (case, rs, int(d.max()), float((d != 0).mean()))
    if rs.persistence > 0.0:
        assert np.abs(state.cpu().numpy().astype(np.float64) - exp_state).max() <= 1e-6


@pytest.mark.parametrize("case", range(24))
def test_random_full_chain_on_the_half_kernel(case):
    """The same draws for float16 frames (BASELINE configs[4]'s pixel format): everything that parks a pre-warp image — a warp or a persistence
    blend behind the chain — runs k_phosphor_ct<R, half> (round 5: qword A phase, raw tile, centre samples in a register window), radii 1 .. 15,
    every aberration, interior / edge / partial strips, a width that is not a multiple of four, a frame size that puts the batch's later frames off
    a qword (33 x 641: the register-window kernel).  Frame by frame against the oracle's float image narrowed as the kernels narrow it
    (|x * 255| in float32, then RNE to half).  No warp: frame 0 is the quantised pre-warp image — bit-exact; elsewhere the 8K test's bar."""
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    rng = np.random.default_rng(9000 + case)
    pick = lambda *a: a[int(rng.integers(len(a)))]
    h, w = CT_SIZES[case % len(CT_SIZES)]
    warp = pick(0.0, 0.15, 0.15, -0.3)
    rs = RenderSettings(
        scanline_strength=pick(0.3, 0.6, 1.0), triad_strength=pick(0.2, 0.35, 0.5, 1.0), triad_gamma=pick(2.2, 1.8, 0.6), triad_preserve_luma=False,
        triad_softness=pick(0.0, 0.5, 1.0, 2.0), aberration_px=int(rng.integers(-8, 9)), bloom_sigma=pick(0.2, 0.5, 1.0, 1.2, 2.0, 2.7, 3.0, 3.7, 4.0, 4.4, 4.6, 5.0),
        bloom_strength=pick(0.1, 0.25, 0.9), bloom_threshold=0.0, noise_strength=pick(0.5, 1.5, 6.0), vignette_strength=pick(0.1, 0.25, 1.0),
        persistence=(pick(0.2, 0.5) if warp == 0.0 else pick(0.0, 0.5)), scanline_speed_px_s=pick(30.0, -12.5, 7.0), scanline_period_px=pick(2.0, 3.7),
        fast_bloom=False, pixel_size=1, warp_strength=warp)
    n, first, fps, seed = 3, int(rng.integers(0, 40)), 25.0, int(rng.integers(1 << 40))
    kind = rng.integers(3)
    frames = (rng.random((n, h, w, 3), dtype=np.float32) * 255.0).astype(np.float16)             # fractional values on the 0..255 scale
    if kind == 1:
        frames[:] = ((np.arange(w, dtype=np.int64)[None, None, :, None] * 7 + np.arange(h)[None, :, None, None] * 3 + np.arange(3)[None, None, None, :] * 50) % 256).astype(np.float16)
    elif kind == 2:
        frames[:, :, : w // 2] = np.float16(255.0)      # saturated half: bloom clips, LUT index 1024
        frames[:, ::5, w // 2:] = np.float16(0.0)
    dev = torch.device("cuda", torch.cuda.current_device())
    pipe = FramePipeline(dev, h, w, rs, fps=fps, noise_seed=seed, dtype=torch.float16)
    out, state = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    want = "k_phosphor_ct<" if (h * w * 6) % 8 == 0 else "k_phosphor_rr<"
    assert pipe.plan().get("phosphor", "").startswith(want) and pipe.plan()["phosphor"].endswith("half>"), pipe.plan()
    got = out.cpu().numpy()
    assert got.dtype == np.float16
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength",
                                          "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "warp_strength")}
    st = None
    for i in range(n):
        p = torch.empty((h, w), dtype=torch.float32, device=dev)
        assert pipe.lib.crtfx_noise_plane(pipe.engine.ctx, seed, first + i, p.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        _, st = orc.process_frames([frames[i]], params, fps, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                   rs.vignette_strength, noise_planes=[p.cpu().numpy()], first_index=first + i, prev_state=st)
        exp16 = np.abs(st.astype(np.float32) * np.float32(255.0)).astype(np.float16)
        if warp == 0.0 and i == 0:
            assert np.array_equal(got[0], exp16), (case, rs)
        diff = np.abs(got[i].astype(np.float32) - exp16.astype(np.float32))
        assert diff.max() <= 0.125 and (got[i] != exp16).mean() < 5e-3, (case, i, rs, float(diff.max()), float((got[i] != exp16).mean()))
        if rs.persistence <= 0.0:
            st = None
    if rs.persistence > 0.0:
        assert np.abs(state.cpu().numpy().astype(np.float64) - st).max() <= 1e-6
