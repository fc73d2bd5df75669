"""GPU parity: pythoncrt_amd (libcrtfx.so through the C-ABI) against the CPU oracle on the same
seeded inputs.  Run on the MI355X box with `pytest -m gpu`.

Bars:
  * everything up to the pre-warp image — normalise, aberration, grade, bloom, triad, scanlines,
    vignette, flicker, injected grain — is compared BIT-EXACTLY (float32(oracle) == gpu, uint8
    equal), except the colour-grade gamma (device powf vs numpy power: <= 3e-7 absolute);
  * the warp's integer sampling map is bit-exact; warped pixels are compared to 3e-7 absolute
    (the reference interpolates its float64 image, the GPU the float32-stored one) and the
    quantised frame to <= 1 LSB with < 0.1 % of samples off.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import crt_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def pc():
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    import pythoncrt_amd
    return pythoncrt_amd


def make_frame(h, w, seed=0, kind="noise"):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    g = np.stack([(xx * 255) // max(1, w - 1), (yy * 255) // max(1, h - 1), ((xx + yy) * 255) // max(1, h + w - 2)], axis=2)
    mix = (g + rng.integers(0, 64, (h, w, 3))) // 2 + 40
    return np.clip(mix, 0, 255).astype(np.uint8)


BASE = dict(scanline_strength=0.0, triad=None, triad_gamma=2.2, triad_preserve_luma=False, aberration_px=0,
            bloom_sigma=0.0, bloom_strength=0.0, bloom_threshold=0.0, noise_strength=0.0, vignette=None,
            scanline_period_px=2.0, scanline_phase_px=0.0, fast_bloom=False, pixel_size=1)


def run_both(pc, frame, cfg, noise_plane=None, **kw):
    """-> (gpu float32 image, oracle float image)."""
    c = dict(BASE, **cfg)
    h, w = frame.shape[:2]
    tm_g = pc.make_triad_mask(h, w, *c["triad"]) if c["triad"] else None
    tm_o = orc.make_triad_mask(h, w, *c["triad"]) if c["triad"] else None
    vg_g = pc.make_vignette(h, w, c["vignette"]) if c["vignette"] else None
    vg_o = orc.make_vignette(h, w, c["vignette"]) if c["vignette"] else None
    pos = lambda tm, vg: (frame, c["scanline_strength"], tm, c["triad_gamma"], c["triad_preserve_luma"], c["aberration_px"],
                          c["bloom_sigma"], c["bloom_strength"], c["bloom_threshold"], c["noise_strength"], vg,
                          c["scanline_period_px"], c["scanline_phase_px"], c["fast_bloom"], c["pixel_size"], 0, 0.0)
    got = pc.apply_static_effects(*pos(tm_g, vg_g), noise_plane=noise_plane, **kw)
    exp = orc.apply_static_effects(*pos(tm_o, vg_o), noise_plane=noise_plane, **kw)
    assert got.dtype == np.float32 and got.shape == frame.shape
    return got, exp


def assert_bit_exact(got, exp):
    e32 = exp.astype(np.float32)
    if not np.array_equal(got, e32):
        d = np.abs(got.astype(np.float64) - e32)
        raise AssertionError(f"{(got != e32).sum()} of {got.size} differ; max |d| = {d.max():.3e}")


SIZES = [(48, 64), (37, 53), (70, 130), (9, 200)]


def test_normalise_exhaustive(pc):
    """a1: every uint8 value / 255.0 (true division) — all effects off."""
    frame = np.arange(256 * 3, dtype=np.uint32).reshape(4, 64, 3) % 256
    frame = frame.astype(np.uint8)
    got, exp = run_both(pc, frame, {})
    assert_bit_exact(got, exp)
    assert set(np.unique(frame)) == set(range(256))


@pytest.mark.parametrize("hw", SIZES)
@pytest.mark.parametrize("d", [-8, -1, 1, 3, 8])
def test_aberration_integer_indexing(pc, hw, d):
    frame = make_frame(*hw, seed=d + 10)
    got, exp = run_both(pc, frame, dict(aberration_px=d))
    assert_bit_exact(got, exp)


@pytest.mark.parametrize("grade", [
    dict(brightness=0.1, contrast=1.2), dict(saturation=1.6), dict(saturation=0.0), dict(temperature=0.7),
    dict(temperature=-1.0), dict(brightness=-0.05, contrast=0.8, saturation=1.3, temperature=-0.4)])
def test_colour_grade_exact(pc, grade):
    frame = make_frame(48, 64, seed=3)
    got, exp = run_both(pc, frame, dict(aberration_px=1), **grade)
    assert_bit_exact(got, exp)


def test_colour_grade_gamma_tolerance(pc):
    frame = make_frame(48, 64, seed=4)
    got, exp = run_both(pc, frame, {}, gamma=2.2, saturation=1.2)
    assert np.abs(got - exp.astype(np.float32)).max() <= 3e-7


@pytest.mark.parametrize("hw", SIZES)
@pytest.mark.parametrize("tri", [
    dict(triad=(0.35, 0.0)), dict(triad=(0.35, 0.5)), dict(triad=(0.35, 0.5), triad_preserve_luma=True),
    dict(triad=(0.8, 1.5), triad_gamma=1.0), dict(triad=(0.5, 0.0), triad_gamma=1.0, triad_preserve_luma=True),
    dict(triad=(0.35, 4.0), triad_gamma=0.5, triad_preserve_luma=True), dict(triad=(1.0, 0.2), triad_gamma=3.3)])
def test_triad_mask(pc, hw, tri):
    frame = make_frame(*hw, seed=5, kind="grad")
    got, exp = run_both(pc, frame, tri)
    assert_bit_exact(got, exp)


def test_triad_mask_as_plain_array(pc):
    """An arbitrary H x W x 3 array (not a descriptor) takes the full-mask path."""
    h, w = 40, 72
    frame = make_frame(h, w, seed=6)
    mask = np.random.default_rng(6).random((h, w, 3), dtype=np.float32)
    a = (frame, 0.0, mask, 2.2, True, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0)
    assert_bit_exact(pc.apply_static_effects(*a), orc.apply_static_effects(*a))
    assert np.array_equal(np.asarray(pc.make_triad_mask(h, w, 0.35, 0.5)), orc.make_triad_mask(h, w, 0.35, 0.5))


@pytest.mark.parametrize("hw", SIZES)
@pytest.mark.parametrize("phase", [0.0, 1.25, 29.0])
def test_scanlines_vignette_flicker(pc, hw, phase):
    frame = make_frame(*hw, seed=7)
    got, exp = run_both(pc, frame, dict(scanline_strength=0.6, scanline_phase_px=phase, vignette=0.25, aberration_px=1,
                                        triad=(0.35, 0.5)),
                        time_sec=0.3, flicker_strength=0.5, flicker_hz=7.0)
    assert exp.dtype == np.float64
    assert_bit_exact(got, exp)
    got, exp = run_both(pc, frame, dict(scanline_strength=0.6, scanline_phase_px=phase))   # config-1 shape
    assert exp.dtype == np.float32
    assert_bit_exact(got, exp)


def test_scanlines_2d_and_vignette_array(pc):
    h, w = 50, 90
    frame = make_frame(h, w, seed=8)
    got, exp = run_both(pc, frame, dict(scanline_strength=0.7, scanline_phase_px=3.0), scanline_angle=7.5, scanline_thickness=1.8)
    assert_bit_exact(got, exp)
    vig = orc.make_vignette(h, w, 0.6)
    a = (frame, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, vig, 2.0, 0.0, False, 1, 0, 0.0)
    assert_bit_exact(pc.apply_static_effects(*a), orc.apply_static_effects(*a))
    assert np.array_equal(np.asarray(pc.make_vignette(h, w, 0.6)), vig)


@pytest.mark.parametrize("hw", SIZES + [(150, 64)])
@pytest.mark.parametrize("sigma", [3.0, 1.2, 0.5, 0.1, 4.0, 4.4, 6.5, 8.3, 10.0, 12.0, 20.0, 27.0, 40.0, 50.0, 85.0])       # radii 9 4 2 1 12 | 13 20 25 30 (one build each) | 36 60 81 120 150 255 (split path)
def test_bloom_bit_exact(pc, hw, sigma):
    """a5: separable Gaussian (LDS strips, H pass, register-window V pass) — same fmaf accumulation order as the
    oracle, so equal to the last bit, borders included.  Radii beyond 30 run the split path (k_sb_rows / k_sb_cols,
    taps from a device array): the same sums at ANY radius, here up to several times the frame size — the reference
    takes any sigma (ref:609-610)."""
    frame = make_frame(*hw, seed=9)
    got, exp = run_both(pc, frame, dict(bloom_sigma=sigma, bloom_strength=0.25, aberration_px=1))
    assert_bit_exact(got, exp)


@pytest.mark.parametrize("hw", [(70, 130), (33, 1030), (9, 200), (1, 1), (3, 5)])
@pytest.mark.parametrize("sigma", [0.5, 3.0, 11.0, 60.0])
def test_split_bloom_any_radius(pc, hw, sigma, monkeypatch):
    """The split path forced for EVERY radius (SPLIT_FROM = 0), full chain with and without the warp, frame widths not a
    multiple of 4 (scalar column pass), rows longer than one 512-px span and a chunked row pass (radius 180 > 256 / 2 steps)."""
    from pythoncrt_amd import effects
    monkeypatch.setattr(effects, "DEBUG_OPTIONS", {"SPLIT_FROM": 0})
    effects._tls.engines = {}
    try:
        h, w = hw
        frame = make_frame(h, w, seed=14, kind="grad")
        plane = np.random.default_rng(14).standard_normal((h, w), dtype=np.float32)
        got, exp = run_both(pc, frame, dict(FULL, bloom_sigma=sigma, bloom_threshold=0.2), noise_plane=plane)
        assert_bit_exact(got, exp)
        got, exp = run_both(pc, frame, dict(FULL, bloom_sigma=sigma), noise_plane=plane, warp_strength=0.15)
        assert np.abs(got.astype(np.float64) - exp).max() <= 3e-7
    finally:
        effects._tls.engines = {}


@pytest.mark.parametrize("sigma", [1.2, 3.0])        # BASELINE configs 2 and 3
def test_gpu_within_every_opencv_variant(pc, sigma):
    """The OpenCV-backed stages are parity-unpinned (no reference-held vector, cv2 not importable).  The oracle restates one
    accumulation form; oracle/crt_oracle.c also restates the OTHER forms OpenCV's C++ engine contains (symmetric-paired
    column taps with / without FMA, SymmRowSmall, multiply-then-add, a contracted remap sum, convertScaleAbs in double).
    Whichever of them a given cv2 build takes, the GPU must stay within the spread tests/test_oracle_variants.py measures:
    float image (no warp) equal to each variant except where a triad-LUT index flips (< 1e-4 of the samples, <= 1e-3);
    uint8 frame with the warp <= 1 LSB on < 0.1 % of the samples (the warp path's own bar)."""
    h, w = 135, 240
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = np.stack([xx * 255.0 / w, yy * 255.0 / h, (xx + yy) * 255.0 / (h + w)], axis=2).astype(np.int32)
    frame = np.clip((rng.integers(0, 256, (h, w, 3), dtype=np.int32) + grad) // 2, 0, 255).astype(np.uint8)
    plane = np.random.default_rng(9).standard_normal((h, w), dtype=np.float32)
    cfg = dict(scanline_strength=0.6, triad=(0.35, 0.5), aberration_px=1, bloom_sigma=sigma, bloom_strength=0.25, vignette=0.25,
               noise_strength=1.5, scanline_phase_px=1.25)
    got, exp0 = run_both(pc, frame, cfg, noise_plane=plane)
    assert_bit_exact(got, exp0)                                   # the oracle's own form: to the bit
    tm_g, vg_g = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
    tm_o, vg_o = orc.make_triad_mask(h, w, 0.35, 0.5), orc.make_vignette(h, w, 0.25)
    a = lambda tm, vg: (frame, 0.6, tm, 2.2, False, 1, sigma, 0.25, 0.0, 1.5, vg, 0.0, None, 2.0, 1.25, False, 1)
    u8_gpu, _ = pc.apply_crt_effect(*a(tm_g, vg_g), warp_strength=0.15, noise_plane=plane)
    for name, kw in orc.OPENCV_VARIANTS.items():
        with orc.opencv_variant(**kw):
            _, exp = run_both(pc, frame, cfg, noise_plane=plane)
            u8_o, _ = orc.apply_crt_effect(*a(tm_o, vg_o), warp_strength=0.15, noise_plane=plane)
        d = np.abs(got.astype(np.float64) - exp.astype(np.float64))
        assert d.max() <= 1e-3 and (d > 1e-6).mean() < 1e-4, (name, float(d.max()), float((d > 1e-6).mean()))
        du = np.abs(np.asarray(u8_gpu).astype(np.int16) - u8_o.astype(np.int16))
        assert du.max() <= 1 and (du != 0).mean() < 1e-3, (name, int(du.max()), float((du != 0).mean()))


def test_bloom_threshold_and_strength(pc):
    frame = make_frame(64, 96, seed=10, kind="grad")
    got, exp = run_both(pc, frame, dict(bloom_sigma=3.0, bloom_strength=1.5, bloom_threshold=0.4), brightness=0.05, contrast=1.1)
    assert_bit_exact(got, exp)
    imp = np.zeros((41, 70, 3), np.uint8)
    imp[20, 35] = 255
    got, exp = run_both(pc, imp, dict(bloom_sigma=3.0, bloom_strength=1.0))
    assert_bit_exact(got, exp)
    assert got[20, 35, 0] == 1.0 and 0 < got[11, 26, 0] < 1e-4 and got[10, 35, 0] == 0.0   # 19x19 support


def test_grain_injected_plane_and_rng(pc):
    h, w = 64, 128
    frame = make_frame(h, w, seed=11, kind="grad")
    plane = np.random.default_rng(11).standard_normal((h, w), dtype=np.float32)
    cfg = dict(noise_strength=8.0, vignette=0.25, scanline_strength=0.6, scanline_phase_px=1.0)
    got, exp = run_both(pc, frame, cfg, noise_plane=plane)
    assert_bit_exact(got, exp)
    got32, exp32 = run_both(pc, frame, dict(noise_strength=8.0), noise_plane=plane)     # float32 tail
    assert exp32.dtype == np.float32
    assert_bit_exact(got32, exp32)
    # in-kernel RNG == the plane crtfx_noise_plane exports for the same (seed, frame)
    from pythoncrt_amd import effects
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = effects._engine(dev, h, w)
    out = torch.empty((h, w), dtype=torch.float32, device=dev)
    rc = eng.lib.crtfx_noise_plane(eng.ctx, 1234, 7, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    gpu_plane = out.cpu().numpy()
    c = dict(BASE, **cfg)
    a = (frame, c["scanline_strength"], None, 2.2, False, 0, 0.0, 0.0, 0.0, c["noise_strength"], pc.make_vignette(h, w, 0.25),
         2.0, 1.0, False, 1, 0, 0.0)
    rng_img = pc.apply_static_effects(*a, noise_seed=1234, frame_index=7)
    plane_img = pc.apply_static_effects(*a, noise_plane=gpu_plane)
    assert np.array_equal(rng_img, plane_img)
    other = pc.apply_static_effects(*a, noise_seed=1234, frame_index=8)
    assert not np.array_equal(rng_img, other)


def test_grain_rng_statistics(pc):
    from pythoncrt_amd import effects
    h, w = 1080, 1920
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = effects._engine(dev, h, w)
    out = torch.empty((h, w), dtype=torch.float32, device=dev)
    eng.lib.crtfx_noise_plane(eng.ctx, 99, 0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    z = out.cpu().numpy().astype(np.float64)
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 0.02 and abs((z ** 4).mean() - 3.0) < 0.05
    assert abs((z[:, 1:] * z[:, :-1]).mean()) < 5 / np.sqrt(n) and abs((z[1:] * z[:-1]).mean()) < 5 / np.sqrt(n)
    from scipy import stats
    assert stats.kstest(z.ravel()[::97], "norm").statistic < 0.01
    out2 = torch.empty_like(out)
    eng.lib.crtfx_noise_plane(eng.ctx, 99, 1, out2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert abs((z * out2.cpu().numpy()).mean()) < 5 / np.sqrt(n)     # frames are independent


@pytest.mark.parametrize("hw,s", [((48, 64), 0.15), ((33, 47), -0.4), ((96, 128), 1.0), ((70, 130), 0.15)])
def test_warp_map_bit_exact(pc, hw, s):
    """Integer tap origins and 5-bit fractions of the barrel map: equal to the oracle's."""
    from pythoncrt_amd import effects
    h, w = hw
    frame = make_frame(h, w, seed=12)
    pc.apply_static_effects(frame, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0, warp_strength=s)
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = effects._engine(dev, h, w)
    ix, iy, fxy = (torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(3))
    rc = eng.lib.crtfx_warp_map(eng.ctx, ix.data_ptr(), iy.data_ptr(), fxy.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    mx, my = orc.barrel_maps(h, w, s)
    oix, oiy, ofxy = orc.remap_quantise(mx, my)
    assert np.array_equal(ix.cpu().numpy(), oix) and np.array_equal(iy.cpu().numpy(), oiy) and np.array_equal(fxy.cpu().numpy(), ofxy)


FULL = dict(scanline_strength=0.6, triad=(0.35, 0.5), aberration_px=1, bloom_sigma=1.2, bloom_strength=0.25,
            noise_strength=1.5, vignette=0.25, scanline_phase_px=1.25)


@pytest.mark.parametrize("hw", [(48, 64), (70, 130), (135, 240)])
@pytest.mark.parametrize("sigma", [1.2, 3.0])
def test_full_chain_with_warp(pc, hw, sigma):
    """BASELINE config 2 / 3 parameter sets at oracle-friendly sizes."""
    h, w = hw
    frame = make_frame(h, w, seed=13, kind="grad")
    plane = np.random.default_rng(13).standard_normal((h, w), dtype=np.float32)
    got, exp = run_both(pc, frame, dict(FULL, bloom_sigma=sigma), noise_plane=plane, warp_strength=0.15)
    assert np.abs(got.astype(np.float64) - exp).max() <= 3e-7
    # without the warp the same chain is bit-exact
    got, exp = run_both(pc, frame, dict(FULL, bloom_sigma=sigma), noise_plane=plane)
    assert_bit_exact(got, exp)


def crt_args(frame, tm, vg, persistence, state, phase, c):
    return (frame, c["scanline_strength"], tm, c["triad_gamma"], c["triad_preserve_luma"], c["aberration_px"], c["bloom_sigma"],
            c["bloom_strength"], c["bloom_threshold"], c["noise_strength"], vg, persistence, state, c["scanline_period_px"], phase,
            c["fast_bloom"], c["pixel_size"])


@pytest.mark.parametrize("warp", [0.0, 0.15])
def test_apply_crt_effect_sequence(pc, warp):
    """Stateful preview path: 4 frames threaded through state_prev (cv2.addWeighted blend)."""
    h, w = 70, 130
    c = dict(BASE, **FULL)
    tm_g, tm_o = pc.make_triad_mask(h, w, 0.35, 0.5), orc.make_triad_mask(h, w, 0.35, 0.5)
    vg_g, vg_o = pc.make_vignette(h, w, 0.25), orc.make_vignette(h, w, 0.25)
    sg = so = None
    for i in range(4):
        frame = make_frame(h, w, seed=20 + i, kind="grad")
        plane = np.random.default_rng(30 + i).standard_normal((h, w), dtype=np.float32)
        ug, sg = pc.apply_crt_effect(*crt_args(frame, tm_g, vg_g, 0.5, sg, float(i), c), warp_strength=warp, noise_plane=plane)
        uo, so = orc.apply_crt_effect(*crt_args(frame, tm_o, vg_o, 0.5, so, float(i), c), warp_strength=warp, noise_plane=plane)
        assert ug.dtype == np.uint8 and sg.dtype == np.float32
        assert np.abs(sg.astype(np.float64) - so).max() <= 4e-7
        d = np.abs(ug.astype(np.int16) - uo.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
        if warp == 0.0 and i == 0:
            assert np.array_equal(ug, uo) and np.array_equal(sg, so.astype(np.float32))


def test_numpy_path_keeps_the_state_on_the_device(pc):
    """The GUI tick's call pattern (ref:1810-1852): numpy frame in, `out, prev = apply_crt_effect(..., state_prev=prev)`.
    The state comes back as a DeviceState — no download, no upload next tick — and the frames and states equal those of
    the same ticks threaded through plain numpy arrays."""
    h, w = 70, 130
    c = dict(BASE, **FULL)
    tm, vg = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
    s_dev = s_np = None
    for i in range(4):
        frame = make_frame(h, w, seed=60 + i, kind="grad")
        kw = dict(warp_strength=0.15, noise_seed=3, frame_index=i)
        u1, s_dev = pc.apply_crt_effect(*crt_args(frame, tm, vg, 0.4, s_dev, float(i), c), **kw)
        u2, s2 = pc.apply_crt_effect(*crt_args(frame, tm, vg, 0.4, s_np, float(i), c), **kw)
        s_np = np.array(s2)                                   # a plain array: uploaded again next tick
        assert isinstance(s_dev, pc.DeviceState) and s_dev._host is None      # still device-resident
        assert isinstance(u1, np.ndarray) and np.array_equal(u1, u2)
        assert np.array_equal(np.asarray(s_dev.tensor.cpu()), s_np)
    assert np.array_equal(np.asarray(s_dev), s_np) and s_dev.shape == (h, w, 3) and s_dev.dtype == np.float32
    # a state of another size (ref:689-690) through the same object
    f2 = make_frame(48, 64, seed=70)
    u3, s3 = pc.apply_crt_effect(*crt_args(f2, pc.make_triad_mask(48, 64, 0.35, 0.5), pc.make_vignette(48, 64, 0.25), 0.4, s_dev, 0.0, c), noise_seed=3, frame_index=9)
    u4, s4 = pc.apply_crt_effect(*crt_args(f2, pc.make_triad_mask(48, 64, 0.35, 0.5), pc.make_vignette(48, 64, 0.25), 0.4, s_np, 0.0, c), noise_seed=3, frame_index=9)
    assert np.array_equal(u3, u4) and np.array_equal(np.asarray(s3), np.asarray(s4))


def test_tensor_in_tensor_out_and_purity(pc):
    h, w = 48, 64
    frame = make_frame(h, w, seed=40)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = torch.from_numpy(frame).to(dev)
    tm, vg = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
    c = dict(BASE, **FULL)
    u, s = pc.apply_crt_effect(*crt_args(t, tm, vg, 0.2, None, 0.0, c), noise_seed=1, frame_index=0)
    assert isinstance(u, torch.Tensor) and u.device == t.device and u.dtype == torch.uint8 and s.dtype == torch.float32
    assert np.array_equal(t.cpu().numpy(), frame)                 # callee never mutates the frame (ref:569 copies)
    s_before = s.clone()
    u2, s2 = pc.apply_crt_effect(*crt_args(t, tm, vg, 0.2, s, 1.0, c), noise_seed=1, frame_index=1)
    assert torch.equal(s, s_before) and not torch.equal(s2, s)   # state_prev is not mutated either
    un, sn = pc.apply_crt_effect(*crt_args(frame, tm, vg, 0.2, None, 0.0, c), noise_seed=1, frame_index=0)
    assert isinstance(un, np.ndarray) and np.array_equal(un, u.cpu().numpy()) and np.array_equal(sn, s.cpu().numpy())


def test_errors_are_loud(pc):
    frame = make_frame(48, 64)
    a = [frame, 0.6, None, 2.2, False, 1, 1.2, 0.25, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0]
    with pytest.raises(ValueError):
        pc.apply_static_effects(frame[:, :, :2], *a[1:])
    with pytest.raises(ValueError):
        pc.apply_static_effects(frame, 0.0, pc.make_triad_mask(10, 10, 0.3), *a[3:])
    with pytest.raises(ValueError):
        pc.apply_static_effects(*a, text_overlay_rgba=np.zeros((10, 10, 3), np.uint8))       # RGBA plane wanted
    pc.apply_static_effects(*a[:6], 40.0, *a[7:])                                            # sigma 40 -> radius 120: runs
    pc.apply_static_effects(*a[:6], 500.0, *a[7:])                                           # sigma 500 -> radius 1500 (31x the frame): runs too
    with pytest.raises(Exception, match="radius"):
        pc.apply_static_effects(*a[:6], 30000.0, *a[7:])                                     # radius 90000: past the tap array's sanity bound


def test_option_values_are_checked(pc, monkeypatch):
    """crtfx_set_option refuses values outside what include/crtfx.h documents (the error text names the limit); a refused switch
    leaves no half-configured ctx behind: the next engine with valid switches works."""
    from pythoncrt_amd import _lib, effects
    frame = make_frame(48, 64)
    a = [frame, 0.6, None, 2.2, False, 1, 1.2, 0.25, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0]
    for bad, pat in (({"WARP_ROWS": 3}, "warp rows"), ({"WARP_ROWS": -1}, "warp rows"), ({"GROUP": 99}, "."), ({"POINT_TILES": 17}, "."), ({"BAND_MB": -2}, "band_mb")):
        monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(bad))
        effects._tls.engines = {}
        with pytest.raises(_lib.CrtfxError, match=pat):
            pc.apply_static_effects(*a)
    monkeypatch.setattr(effects, "DEBUG_OPTIONS", {"WARP_ROWS": 0, "GROUP": 2})
    effects._tls.engines = {}
    pc.apply_static_effects(*a, warp_strength=0.15)
    effects._tls.engines = {}


# ---- BASELINE full sizes: size-independent properties ------------------------------------------

@pytest.mark.parametrize("hw", [(1080, 1920), (2160, 3840)])
def test_full_size_properties(pc, hw):
    h, w = hw
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(1234)
    frame = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, generator=g)
    fd = frame.to(dev)
    off = (fd, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 0.0, None, 2.0, 0.0, False, 1)
    u, s = pc.apply_crt_effect(*off)
    assert torch.equal(u, fd)                                   # identity chain: u8 -> /255 -> *255 round trip
    assert np.array_equal(s.cpu().numpy(), frame.numpy().astype(np.float32) / 255.0)   # true division (torch's GPU x/255 is x*(1/255))
    # aberration only: an exact wrap-around shift of R and B (integer indexing)
    a = list(off)
    a[5] = 3
    u, _ = pc.apply_crt_effect(*a)
    exp = torch.stack([torch.roll(fd[:, :, 0], 3, 1), fd[:, :, 1], torch.roll(fd[:, :, 2], -3, 1)], dim=2)
    assert torch.equal(u, exp)
    # bloom of a constant image is that constant (taps sum to 1 within float rounding); borders replicate
    const = torch.full((h, w, 3), 100, dtype=torch.uint8, device=dev)
    b = list(off)
    b[0], b[6], b[7] = const, 3.0, 0.5
    img = pc.apply_static_effects(*b[:11], *b[13:], 0, 0.0)
    v = 100 / 255
    assert float((img - (v + 0.5 * v)).abs().max()) < 2e-6
    # full chain, config-3 parameters: deterministic for a fixed (seed, frame); finite; in range;
    # and equal to the oracle on a window cut from the middle of the frame
    tm, vg = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
    c = dict(BASE, **dict(FULL, bloom_sigma=3.0))
    u1, s1 = pc.apply_crt_effect(*crt_args(fd, tm, vg, 0.0, None, 1.25, c), warp_strength=0.15, noise_seed=5, frame_index=2)
    u2, s2 = pc.apply_crt_effect(*crt_args(fd, tm, vg, 0.0, None, 1.25, c), warp_strength=0.15, noise_seed=5, frame_index=2)
    assert torch.equal(u1, u2) and torch.equal(s1, s2)
    assert bool(torch.isfinite(s1).all()) and float(s1.min()) >= 0.0 and float(s1.max()) <= 1.0
    assert not bool(u1[0, 0].any()) and not bool(u1[-1, -1].any())   # barrel warp: corners sample outside -> 0


def test_1080p_frame_against_oracle(pc):
    """One whole 1080p frame, BASELINE config 2 parameters, against the oracle (a few seconds of CPU)."""
    h, w = 1080, 1920
    frame = make_frame(h, w, seed=50, kind="grad")
    plane = np.random.default_rng(50).standard_normal((h, w), dtype=np.float32)
    got, exp = run_both(pc, frame, FULL, noise_plane=plane)
    assert_bit_exact(got, exp)
    gotw, expw = run_both(pc, frame, FULL, noise_plane=plane, warp_strength=0.15)
    assert np.abs(gotw.astype(np.float64) - expw).max() <= 3e-7
    d = np.abs(orc.convert_scale_abs(gotw).astype(np.int16) - orc.convert_scale_abs(expw).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3


def test_kernel_variants_agree(pc, monkeypatch):
    """The four k_phosphor builds of one launch — the column-owner kernel k_phosphor_cc, the gate-folded
    register-window kernel, its runtime-flag instantiation, and the generic LDS-ring kernel — give identical bits."""
    from pythoncrt_amd import effects
    h, w = 150, 200
    frame = make_frame(h, w, seed=60, kind="grad")
    tm, vg = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
    outs = {}
    for name, opts in (("folded", {}), ("cc", {"FORCE_CC": 1}), ("cc_no_ct", {"FORCE_CC": 1, "NO_CT": 1}), ("no_cc", {"NO_CC": 1}), ("runtime_flags", {"FORCE_RUNTIME_FLAGS": 1}), ("generic", {"FORCE_GENERIC": 1}),
                       ("split", {"SPLIT_FROM": 0}), ("split_plane", {"SPLIT_FROM": 0, "SPLIT_SRC_PLANE": 1}),
                       ("warp_rows_1", {"WARP_ROWS": 1}), ("warp_rows_2", {"WARP_ROWS": 2}), ("warp_rows_4", {"WARP_ROWS": 4}),      # k_warp_lean's tile shapes (0 = the launcher's choice)
                       ("no_plain_warp", {"NO_PLAIN_WARP": 1})):      # ... and its general build where the branch-free one is the default
        monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(opts))
        effects._tls.engines = {}          # the switches are applied when a ctx is created
        res = []
        for sigma in (3.0, 1.2, 2.0, 4.4, 5.0, 5.5, 10.0):      # radii 9, 4, 6, 13, 15 (k_phosphor_ct; from 13 with a few spilled registers), 16, 30 (k_phosphor_cc)
            a = (frame, 0.6, tm, 2.2, False, 1, sigma, 0.25, 0.0, 1.5, vg, 2.0, 1.25, False, 1, 0, 0.0)
            res.append(pc.apply_static_effects(*a, noise_seed=7, frame_index=3))
            res.append(pc.apply_static_effects(*a, noise_seed=7, frame_index=3, warp_strength=0.15))
        # the render loop (uint8 out, persistence state): lean k_point / k_warp builds against the general ones
        from pythoncrt_amd.pipeline import FramePipeline, RenderSettings, baseline_config
        dev = torch.device("cuda", torch.cuda.current_device())
        clip4 = torch.from_numpy(np.stack([make_frame(h, w, seed=61 + i, kind="grad") for i in range(4)])).to(dev)
        for rs in (RenderSettings(), RenderSettings(pixel_size=1, persistence=0.0), RenderSettings(warp_strength=0.2),
                   baseline_config(2)[0], baseline_config(4)[0],
                   RenderSettings(fast_bloom=False, bloom_sigma=2.0, scanline_angle=12.0, scanline_thickness=2.0, grain_size=2, warp_strength=0.15),
                   RenderSettings(fast_bloom=False, bloom_sigma=5.0, grain_size=3, pixel_size=1, persistence=0.0),
                   RenderSettings(fast_bloom=False, bloom_sigma=2.0), RenderSettings(fast_bloom=False, bloom_sigma=3.0, pixel_size=1)):      # Gaussian + persistence, no warp
            pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=5)
            keep = torch.empty((4, h, w, 3), dtype=torch.float32, device=dev) if rs.persistence > 0 else None
            o, st = pipe.run(clip4, first_index=2, local_states=keep)
            res.append(o.cpu().numpy())
            if st is not None:
                res.append(st.cpu().numpy())
            if keep is not None:
                res.append(keep.cpu().numpy())      # the per-frame states the sharded render's fix-up reads (written even when a run keeps its state in registers)
        outs[name] = res
    effects._tls.engines = {}
    for name in ("cc", "cc_no_ct", "no_cc", "runtime_flags", "generic", "split", "split_plane", "warp_rows_1", "warp_rows_2", "warp_rows_4", "no_plain_warp"):
        for x, y in zip(outs["folded"], outs[name]):
            assert np.array_equal(x, y), name


@pytest.mark.parametrize("hw", [(72, 128), (150, 200), (34, 66), (16, 64), (270, 480), (2, 2), (18, 1000)])
def test_fused_fast_bloom_kernel_equals_the_two_launch_path(pc, hw, monkeypatch):
    """k_point_fused_seq (round 6: the half-resolution fast-bloom source of ref:605-607 formed in LDS inside the pointwise kernel) against
    k_half_group + k_point_lean_seq (the source as a plane): identical bytes and states — every tile position (first / last column strip, a last
    block row that hangs over the frame, frames smaller than a tile), pixel sizes 1 / 2 / 3, aberration, with and
    without persistence, a warp behind it, half frames, and blocks of 4 / 8 / 16 wavefronts.  Frames of odd size take the two-launch path either way."""
    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    h, w = hw
    dev = torch.device("cuda", torch.cuda.current_device())
    n = 11
    clip_u8 = np.stack([make_frame(h, w, seed=300 + i, kind="grad" if i % 2 else "noise") for i in range(n)])
    # the gate-folded builds (the reference CLI's default gate set with or without pixelate) ...
    cases = [RenderSettings(), RenderSettings(pixel_size=1), RenderSettings(pixel_size=3, aberration_px=3), RenderSettings(persistence=0.0),
             RenderSettings(pixel_size=1, persistence=0.0, aberration_px=0, bloom_strength=0.9),
             RenderSettings(warp_strength=0.2), RenderSettings(pixel_size=1, warp_strength=0.15, persistence=0.0, bloom_strength=0.6),
             # ... and the run-time-gate builds (SF_LEAN_RT): one knob away from the defaults — a colour grade (table and arithmetic forms), a bloom
             # threshold, stages switched off (no vignette: the chain stays float32), flicker, preserve-luma
             RenderSettings(brightness=0.05, contrast=1.1), RenderSettings(saturation=1.3, gamma=1.8, temperature=0.2, pixel_size=1),
             RenderSettings(bloom_threshold=0.3, persistence=0.0), RenderSettings(vignette_strength=0.0, noise_strength=0.0),
             RenderSettings(triad_strength=0.0, scanline_strength=0.0, pixel_size=3), RenderSettings(flicker_strength=0.2, flicker_hz=50.0, triad_preserve_luma=True),
             RenderSettings(vignette_strength=0.0, persistence=0.0, gamma=0.8, warp_strength=0.1),
             # ... and ONE knob the grade table cannot express: folded builds of their own for uint8 frames (+sat / -grain / -vignette / +flicker here)
             RenderSettings(saturation=1.2), RenderSettings(noise_strength=0.0), RenderSettings(vignette_strength=0.0, pixel_size=1),
             RenderSettings(flicker_strength=0.1, flicker_hz=50.0), RenderSettings(grain_size=2), RenderSettings(grain_size=3, pixel_size=1, persistence=0.0),
             RenderSettings(scanline_angle=10.0), RenderSettings(scanline_thickness=2.0, scanline_angle=-4.0, pixel_size=1, persistence=0.0),      # a 2-D scanline plane per frame
             RenderSettings(bloom_strength=0.0)]
    outs = {}
    for name, opts in (("fused", {}), ("two", {"NO_FUSED_HALF": 1}), ("fused4", {"POINT_TILES": 4}), ("fused16", {"POINT_TILES": 16}),
                       ("general", {"FORCE_RUNTIME_FLAGS": 1})):      # k_half_group<runtime> + k_point_sel_seq: the kernels every gate set ran on before round 6
        monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(opts))
        effects._tls.engines = {}
        res, plans = [], []
        for dtype in (torch.uint8, torch.float16):
            frames = torch.from_numpy(clip_u8).to(dev).to(dtype)
            for rs in cases:
                pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=9, dtype=dtype)
                o, st = pipe.run(frames, first_index=3)
                if rs.warp_strength == 0.0:             # (with a warp behind it the launcher may take the frames one by one: k_point_lean)
                    plans.append(pipe.plan().get("point", ""))
                res.append(o.cpu().numpy())
                if st is not None:
                    res.append(st.cpu().numpy())
                    o2, st2 = pipe.run(frames[:5], first_index=3 + n, state=st)          # a batch that continues from a carried state
                    res += [o2.cpu().numpy(), st2.cpu().numpy()]
        outs[name] = (res, plans)
    effects._tls.engines = {}
    even = h % 2 == 0 and w % 2 == 0
    # (the last case has no bloom: nothing to fuse — a folded k_point_lean_seq for uint8 frames, the run-time form for half frames)
    # (coarse grain: a folded build of the fused kernel for uint8 frames; half frames stay on the general k_point_sel_seq)
    ok = ("k_point_fused_seq<", "k_point_lean_seq<fast+pixelate-bloom,u8", "k_point_lean_seq<runtime,half", "k_point_sel_seq<half")
    assert not even or all(p.startswith(ok) for p in outs["fused"][1]), outs["fused"][1]
    assert any(p.startswith("k_point_fused_seq<") for p in outs["fused"][1]) == even
    assert not any(p.startswith("k_point_fused_seq<") for p in outs["two"][1]), outs["two"][1]
    assert not any(p.startswith(("k_point_fused_seq<", "k_point_lean_seq<")) for p in outs["general"][1]), outs["general"][1]
    if even:
        # (first) six plane-free, warp-free gate sets x two pixel formats: four keep the defaults' loads (folded) — a bloom threshold alone stays on the
        # fully folded build (that bit acts on the bloom source only and is read at run time there); uint8 frames with a per-channel grade read
        # it from the host's table (+gradelut: 1 case); a saturation change, flicker + preserve-luma, and every grade of half frames run it at
        # run time (+grade: 2 + 3) — and two switch stages off (the gate word wholly at run time)
        names = outs["fused"][1]
        assert sum("+gradelut," in p for p in names) == 1 and sum("+grade," in p for p in names) == 2 + 3 + 2, names
        assert sum(p.startswith("k_point_fused_seq<runtime") for p in names) == 2 * 2 + 2, names
        # the one-knob cases: a folded build for uint8 frames, a run-time form for half frames
        assert [sum(k in p for p in names) for k in ("+pixelate+sat,u8", "+pixelate-grain,u8", "fast-vignette,u8", "+pixelate+flicker,u8")] == [1, 1, 1, 1], names
        assert [sum(k in p for p in names) for k in ("fast+pixelate+coarse,u8,render", "fast+coarse,u8,none")] == [1, 1], names
        assert [sum(k in p for p in names) for k in ("fast+pixelate+scan2d,u8,render", "fast+scan2d,u8,none")] == [1, 1], names
    for name in ("two", "fused4", "fused16", "general"):
        assert len(outs[name][0]) == len(outs["fused"][0])
        for k, (x, y) in enumerate(zip(outs["fused"][0], outs[name][0])):
            assert np.array_equal(x, y), (name, k, hw)


def test_fused_fast_bloom_odd_sizes_take_the_plane_path(pc):
    """cv2.resize's exact 2x decimation only exists for even sizes: an odd width or height keeps the generic bilinear taps, i.e. the k_half_group plane."""
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    dev = torch.device("cuda", torch.cuda.current_device())
    for h, w in ((71, 128), (72, 127)):
        pipe = FramePipeline(dev, h, w, RenderSettings(), fps=30.0, noise_seed=9)
        pipe.run(torch.zeros((4, h, w, 3), dtype=torch.uint8, device=dev))
        assert pipe.plan().get("point", "").startswith("k_point_lean_seq<") and pipe.plan().get("half", "").startswith("k_half_group<"), pipe.plan()


@pytest.mark.parametrize("triad", [(0.35, 0.5), (0.35, 0.0), (0.5, 1.0), (0.2, 2.0), (1.0, 0.7)])
@pytest.mark.parametrize("hw", [(40, 700), (90, 130), (33, 64), (20, 1)])
def test_composite_triad_tables(pc, triad, hw, monkeypatch):
    """k_phosphor_ct gathers lut_inv[idx(lut_g[i] * m)] from ONE table per mask value where a strip's mask has at most the two
    tabulated values, and runs the two-LUT form elsewhere (border columns of a softened mask, a three-valued mask): interior
    strips, edge strips, partial last strips and masks of 2 / 3 / many values against the oracle (bit-exact: ref:246-263) and
    against k_phosphor_cc, for pre-warp launches (warp on) of several radii."""
    from pythoncrt_amd import effects
    h, w = hw
    frame = make_frame(h, w, seed=77, kind="grad")
    got = {}
    for name, opts in (("ct", {"FORCE_CC": 1}), ("cc", {"FORCE_CC": 1, "NO_CT": 1})):
        monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(opts))
        effects._tls.engines = {}
        res = []
        for sigma in (3.0, 1.2, 4.0):
            c = dict(BASE, **dict(FULL, triad=triad, bloom_sigma=sigma))
            tm = pc.make_triad_mask(h, w, *triad)
            vg = pc.make_vignette(h, w, c["vignette"])
            a = (frame, c["scanline_strength"], tm, 2.2, False, c["aberration_px"], sigma, c["bloom_strength"], 0.0, c["noise_strength"], vg,
                 2.0, 1.25, False, 1, 0, 0.0)
            res.append(pc.apply_static_effects(*a, noise_seed=11, frame_index=2, warp_strength=0.15))      # pre-warp image parked -> cc / ct
        got[name] = res
    effects._tls.engines = {}
    for x, y in zip(got["ct"], got["cc"]):
        assert np.array_equal(x, y)
    # and the no-warp float image of the same chain against the oracle, bit for bit (the pre-warp image itself)
    monkeypatch.setattr(effects, "DEBUG_OPTIONS", {"FORCE_CC": 1})
    effects._tls.engines = {}
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    dev = torch.device("cuda", torch.cuda.current_device())
    rs = RenderSettings(fast_bloom=False, bloom_sigma=3.0, pixel_size=1, persistence=0.5, triad_strength=triad[0], triad_softness=triad[1])     # persistence: the chain parks a pre-warp image
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=5)
    clip = np.stack([make_frame(h, w, seed=79 + i, kind="grad") for i in range(3)])
    out, _ = pipe.run(torch.from_numpy(clip).to(dev), first_index=1)
    gpl = _export_planes(pipe, 5, 1, 3, h, w)
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength",
                                          "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "warp_strength")}
    exp, _ = orc.process_frames(list(clip), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                rs.vignette_strength, noise_planes=gpl, first_index=1)
    assert np.array_equal(out[0].cpu().numpy(), exp[0])            # frame 0 passes through unblended: the pre-warp image quantised, bit-exact
    d = np.abs(out.cpu().numpy().astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
    effects._tls.engines = {}


@pytest.mark.parametrize("ab", [0, 2, -1, -3, 5, 8, -8])
@pytest.mark.parametrize("w,sigma", [(700, 3.0), (1028, 3.0), (700, 5.0), (1028, 0.4)])      # radii 9, 15 (the widest window), 1
def test_ct_frame_row_windows(pc, ab, w, sigma, monkeypatch):
    """k_phosphor_ct's dword A phase reads each strip's staged row segment as ONE window of the frame row, R and B displaced by the
    aberration (ref:571-577): every shift the CLI admits (-8 .. 8, ref:1230) and none, on frames wide enough for interior strips
    (window inside the frame) next to edge strips (byte-wise path: BORDER_REPLICATE + wrap), against k_phosphor_cc bit for bit and
    against the oracle through the render loop (frame 0 bit-exact)."""
    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    h = 26
    dev = torch.device("cuda", torch.cuda.current_device())
    clip = np.stack([make_frame(h, w, seed=90 + i, kind="noise" if i else "grad") for i in range(2)])
    rs = RenderSettings(fast_bloom=False, bloom_sigma=sigma, pixel_size=1, persistence=0.5, aberration_px=ab)     # persistence: the chain parks a pre-warp image
    got = {}
    for name, opts in (("ct", {"FORCE_CC": 1}), ("cc", {"FORCE_CC": 1, "NO_CT": 1})):
        monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(opts))
        effects._tls.engines = {}
        pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=5)
        out, st = pipe.run(torch.from_numpy(clip).to(dev), first_index=1)
        got[name] = (out.cpu().numpy(), st.cpu().numpy())
        if name == "ct":
            gpl = _export_planes(pipe, 5, 1, 2, h, w)
    effects._tls.engines = {}
    assert np.array_equal(got["ct"][0], got["cc"][0]) and np.array_equal(got["ct"][1], got["cc"][1])
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength",
                                          "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "warp_strength")}
    exp, _ = orc.process_frames(list(clip), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                rs.vignette_strength, noise_planes=gpl, first_index=1)
    assert np.array_equal(got["ct"][0][0], exp[0])
    d = np.abs(got["ct"][0].astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3


def test_sharded_persistence_pieces_on_gpu():
    """The GPU engine behind shard.ShardedRender: a chunk scanned from a ZERO incoming state and then
    corrected with p^(j+1) * carry reproduces the in-order render of the same frames."""
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from pythoncrt_amd.pipeline import FramePipeline, GpuShardEngine, baseline_config
    dev = torch.device("cuda", torch.cuda.current_device())
    rs, _, _ = baseline_config(4)                 # 1080p config-4 parameters (persistence 0.5) at a small size
    h, w, B = 135, 240, 6
    g = torch.Generator(device="cpu").manual_seed(3)
    frames = torch.randint(0, 256, (2 * B, h, w, 3), dtype=torch.uint8, generator=g).to(dev)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=11)
    # in-order reference on the GPU: 2B frames, state threaded through; keep every state
    states = torch.empty((2 * B, h, w, 3), dtype=torch.float32, device=dev)
    seq_out, _ = pipe.run(frames, first_index=0, local_states=states)
    # chunk 1 as a remote rank would do it
    eng = GpuShardEngine(pipe, B)
    local, out = eng.local_scan(frames[B:], first_index=B, clip_start=False)
    carry = states[B - 1].clone()
    eng.correct(local[:B], carry, rs.persistence, out)
    torch.cuda.synchronize()
    d = (out.cpu().to(torch.int16) - seq_out[B:].cpu().to(torch.int16)).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-3
    true_states = local[:B] + torch.tensor([rs.persistence ** (j + 1) for j in range(B)], device=dev).view(B, 1, 1, 1) * carry
    assert float((true_states - states[B:]).abs().max()) < 5e-7
    assert torch.equal(local[B - 1], local[:B][B - 1])           # the chunk-final state (what travels to the next rank)
    # chunk 0 (clip start) needs no carry: identical to the in-order frames
    local0, out0 = eng.local_scan(frames[:B], first_index=0, clip_start=True)
    torch.cuda.synchronize()
    assert torch.equal(out0, seq_out[:B]) and torch.equal(local0[:B], states[:B])
    # a chunk LONGER than the settling time (p = 0.5: 26 frames): per-frame states only for its first 26 frames, the rest of
    # the chunk runs with the state in registers; frames, kept states and the chunk-final state still those of the in-order render
    from pythoncrt_amd.shard import settle_frames
    BL, hs, ws = 40, 48, 64
    K = settle_frames(rs.persistence, 2.0 ** -26)
    assert K < BL
    fl = torch.randint(0, 256, (2 * BL, hs, ws, 3), dtype=torch.uint8, generator=g).to(dev)
    pl = FramePipeline(dev, hs, ws, rs, fps=30.0, noise_seed=12)
    sl = torch.empty((2 * BL, hs, ws, 3), dtype=torch.float32, device=dev)
    seq_l, _ = pl.run(fl, first_index=0, local_states=sl)
    el = GpuShardEngine(pl, BL, slots=2)
    assert el.keep == K and el.local[0].shape[0] == K
    loc0, o0 = el.local_scan(fl[:BL], first_index=0, clip_start=True, slot=0)
    loc1, o1 = el.local_scan(fl[BL:], first_index=BL, clip_start=False, slot=1)
    el.correct(loc1[:K], sl[BL - 1].clone(), rs.persistence, o1[:K])
    torch.cuda.synchronize()
    assert torch.equal(o0, seq_l[:BL]) and torch.equal(loc0[:K], sl[:K]) and torch.equal(loc0[BL - 1], sl[BL - 1])
    d = (o1.cpu().to(torch.int16) - seq_l[BL:].cpu().to(torch.int16)).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-3
    assert float((loc1[BL - 1] + (rs.persistence ** BL) * sl[BL - 1] - sl[2 * BL - 1]).abs().max()) < 5e-7
    with pytest.raises(IndexError):
        loc1[:K + 1]
    # one rank: ShardedRender carries the state itself from chunk to chunk — the in-order frames, bit for bit
    from pythoncrt_amd.shard import FrameShard, ShardedRender
    render = ShardedRender(FrameShard(1, 0, B), rs.persistence, GpuShardEngine(pipe, B), dist=None)
    got = torch.cat([render.run_round(frames[r * B:(r + 1) * B], r).clone() for r in range(2)])
    assert torch.equal(got, seq_out)
    # the carried state is the last frame's, and the batch call leaves it in state_inout
    st = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    keep = torch.empty((B, h, w, 3), dtype=torch.float32, device=dev)
    _, st_out = pipe.run(frames[:B], first_index=0, state=None, local_states=keep)
    assert torch.equal(st_out, keep[B - 1]) and torch.equal(keep, states[:B])


# ---- pixelate, text overlay, glitch, grain size, fast bloom ----------------------------

def static_args(frame, tm, vg, c, glitch=(0, 0.0)):
    return (frame, c["scanline_strength"], tm, c["triad_gamma"], c["triad_preserve_luma"], c["aberration_px"], c["bloom_sigma"],
            c["bloom_strength"], c["bloom_threshold"], c["noise_strength"], vg, c["scanline_period_px"], c["scanline_phase_px"],
            c["fast_bloom"], c["pixel_size"], glitch[0], glitch[1])


def both_static(pc, frame, cfg, glitch=(0, 0.0), **kw):
    c = dict(BASE, **cfg)
    h, w = frame.shape[:2]
    tm_g = pc.make_triad_mask(h, w, *c["triad"]) if c["triad"] else None
    tm_o = orc.make_triad_mask(h, w, *c["triad"]) if c["triad"] else None
    vg_g = pc.make_vignette(h, w, c["vignette"]) if c["vignette"] else None
    vg_o = orc.make_vignette(h, w, c["vignette"]) if c["vignette"] else None
    return (pc.apply_static_effects(*static_args(frame, tm_g, vg_g, c, glitch), **kw),
            orc.apply_static_effects(*static_args(frame, tm_o, vg_o, c, glitch), **kw))


@pytest.mark.parametrize("hw", [(48, 64), (37, 53), (70, 130)])
@pytest.mark.parametrize("px", [2, 3, 7])
def test_pixelate(pc, hw, px):
    """a3: the INTER_NEAREST down/up pair as composite index maps — integer indexing, bit-exact; also through
    the bloom kernels (the lean one reads its row map from LDS)."""
    frame = make_frame(*hw, seed=70)
    got, exp = both_static(pc, frame, dict(pixel_size=px, aberration_px=1))
    assert_bit_exact(got, exp)
    got, exp = both_static(pc, frame, dict(pixel_size=px, aberration_px=-2, bloom_sigma=3.0, bloom_strength=0.25,
                                           scanline_strength=0.6, scanline_phase_px=1.0, triad=(0.35, 0.5), vignette=0.25))
    assert_bit_exact(got, exp)


def make_overlay(h, w, seed):
    rng = np.random.default_rng(seed)
    ov = np.zeros((h, w, 4), np.uint8)
    ov[h // 4: h // 2, w // 8: w // 2] = rng.integers(0, 256, (h // 2 - h // 4, w // 2 - w // 8, 4), dtype=np.uint8)
    ov[h // 2:, :, 3] = 255
    ov[h // 2:, :, :3] = rng.integers(0, 256, (h - h // 2, w, 3), dtype=np.uint8)
    return ov


@pytest.mark.parametrize("after", [False, True])
@pytest.mark.parametrize("cfg", [dict(scanline_strength=0.6, scanline_phase_px=2.0, aberration_px=1),
                                 dict(FULL, bloom_sigma=3.0, noise_strength=0.0),
                                 dict(FULL, bloom_sigma=1.2, noise_strength=0.0, vignette=None)])
def test_text_overlay(pc, after, cfg):
    """8f row 1: alpha blend of a supplied RGBA plane before (enters the bloom) or after the effects."""
    h, w = 70, 130
    frame = make_frame(h, w, seed=71, kind="grad")
    ov = make_overlay(h, w, 72)
    got, exp = both_static(pc, frame, cfg, text_overlay_rgba=ov, text_overlay_after=after)
    assert_bit_exact(got, exp)
    gotw, expw = both_static(pc, frame, cfg, text_overlay_rgba=ov, text_overlay_after=after, warp_strength=0.15)
    assert np.abs(gotw.astype(np.float64) - expw).max() <= 3e-7


@pytest.mark.parametrize("warp", [0.0, 0.15])
def test_glitch_render_and_preview(pc, warp):
    """8f row 2: both glitch variants (offsets from numpy's PCG64 with the reference's seeds, gather on the GPU)."""
    h, w = 96, 160
    frame = make_frame(h, w, seed=73, kind="grad")
    cfg = dict(scanline_strength=0.6, scanline_phase_px=13.0, aberration_px=1, triad=(0.5, 0.0), bloom_sigma=1.2, bloom_strength=0.25)
    got, exp = both_static(pc, frame, cfg, glitch=(9, 0.4), warp_strength=warp, text_overlay_rgba=make_overlay(h, w, 74))
    if warp == 0.0:
        assert_bit_exact(got, exp)
    else:
        assert np.abs(got.astype(np.float64) - exp).max() <= 3e-7
    c = dict(BASE, **cfg)
    tm_g, tm_o = pc.make_triad_mask(h, w, 0.5, 0.0), orc.make_triad_mask(h, w, 0.5, 0.0)
    ug, sg = pc.apply_crt_effect(*crt_args(frame, tm_g, None, 0.0, None, 250.0, c), glitch_amp_px=11, glitch_height_frac=0.5, warp_strength=warp)
    uo, so = orc.apply_crt_effect(*crt_args(frame, tm_o, None, 0.0, None, 250.0, c), glitch_amp_px=11, glitch_height_frac=0.5, warp_strength=warp)
    if warp == 0.0:
        assert np.array_equal(ug, uo) and np.array_equal(sg, so.astype(np.float32))
    else:
        assert np.abs(sg.astype(np.float64) - so).max() <= 3e-7 and np.abs(ug.astype(np.int16) - uo.astype(np.int16)).max() <= 1


@pytest.mark.parametrize("hw,g", [((48, 64), 2), ((37, 53), 3), ((70, 130), 8), ((20, 30), 64)])
def test_grain_size(pc, hw, g):
    """8f row 3: grain drawn at (H//g, W//g) and bilinearly upsampled (cv2.resize INTER_LINEAR)."""
    h, w = hw
    frame = make_frame(h, w, seed=75, kind="grad")
    plane = np.random.default_rng(76).standard_normal((max(1, h // g), max(1, w // g)), dtype=np.float32)
    cfg = dict(noise_strength=8.0, vignette=0.25, bloom_sigma=1.2, bloom_strength=0.25)
    got, exp = both_static(pc, frame, cfg, grain_size=g, noise_plane=plane)
    assert_bit_exact(got, exp)
    a = pc.apply_static_effects(*static_args(frame, None, None, dict(BASE, noise_strength=8.0)), grain_size=g, noise_seed=3, frame_index=1)
    b = pc.apply_static_effects(*static_args(frame, None, None, dict(BASE, noise_strength=8.0)), grain_size=g, noise_seed=3, frame_index=1)
    assert np.array_equal(a, b) and a.std() > 0


@pytest.mark.parametrize("hw", [(48, 64), (70, 130), (37, 53), (64, 51), (135, 240)])
def test_fast_bloom(pc, hw):
    """8f row 3: the CLI-default bloom — half-res bilinear down (2x2 mean when exact) and up."""
    frame = make_frame(*hw, seed=77, kind="grad")
    got, exp = both_static(pc, frame, dict(fast_bloom=True, bloom_sigma=1.2, bloom_strength=0.25, aberration_px=1))
    assert_bit_exact(got, exp)
    got, exp = both_static(pc, frame, dict(FULL, fast_bloom=True, bloom_threshold=0.3, noise_strength=0.0, bloom_sigma=0.0))
    assert_bit_exact(got, exp)


def test_cli_default_settings_chain(pc):
    """The reference CLI's default flag set (ref:1160-1206: pixel_size 2, fast bloom, persistence 0.2, ...)
    through the frame pipeline against the oracle's in-order render."""
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    h, w, n = 90, 160, 5
    dev = torch.device("cuda", torch.cuda.current_device())
    rs = RenderSettings()                 # CLI defaults
    frames = [make_frame(h, w, seed=80 + i, kind="grad") for i in range(n)]
    planes = np.random.default_rng(81).standard_normal((n, h, w), dtype=np.float32)
    pipe = FramePipeline(dev, h, w, rs, fps=24.0, noise_seed=1)
    out, _ = pipe.run(torch.from_numpy(np.stack(frames)).to(dev), noise_planes=torch.from_numpy(planes).to(dev))
    params = dict(scanline_strength=rs.scanline_strength, triad_gamma=rs.triad_gamma, triad_preserve_luma=rs.triad_preserve_luma,
                  aberration_px=rs.aberration_px, bloom_sigma=rs.bloom_sigma, bloom_strength=rs.bloom_strength, noise_strength=rs.noise_strength,
                  scanline_period_px=rs.scanline_period_px, fast_bloom=rs.fast_bloom, pixel_size=rs.pixel_size)
    exp, _ = orc.process_frames(frames, params, 24.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                rs.vignette_strength, noise_planes=list(planes))
    d = np.abs(out.cpu().numpy().astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
    assert np.array_equal(out[0].cpu().numpy(), exp[0])     # first frame: no blend yet -> bit-exact


# ---- fp16 pixels (BASELINE config 5) -----------------------------------------------------------------------

def f16_frame(h, w, seed):
    rng = np.random.default_rng(seed)
    return (rng.random((h, w, 3), dtype=np.float32) * 255.0).astype(np.float16)      # fractional values on the 0..255 scale


@pytest.mark.parametrize("cfg", [dict(aberration_px=1), dict(FULL, bloom_sigma=3.0), dict(FULL, bloom_sigma=1.2, triad_preserve_luma=True),
                                 dict(FULL, bloom_sigma=6.5), dict(FULL, fast_bloom=True, pixel_size=2),
                                 dict(FULL, noise_strength=0.0, bloom_sigma=2.0, bloom_threshold=0.2),     # runtime-gate half build of k_phosphor_rr
                                 dict(FULL, noise_strength=0.0, bloom_sigma=5.0, triad_preserve_luma=True, pixel_size=2),
                                 dict(FULL, bloom_sigma=11.0), dict(FULL, bloom_sigma=30.0, pixel_size=2, bloom_threshold=0.3)])      # radii 33, 90: the split path on half frames
def test_fp16_frames(pc, cfg):
    """A float16 frame is the reference's frame array held as half (ref:569 divides whatever it gets by
    255.0); the float image is compared as for uint8 frames, the half output frame is |x*255| narrowed."""
    h, w = 70, 130
    frame = f16_frame(h, w, 90)
    plane = np.random.default_rng(91).standard_normal((h, w), dtype=np.float32)
    got, exp = both_static(pc, frame, cfg, noise_plane=plane if dict(BASE, **cfg)["noise_strength"] > 0 else None)
    assert_bit_exact(got, exp)
    c = dict(BASE, **cfg)
    tm_g = pc.make_triad_mask(h, w, *c["triad"]) if c["triad"] else None
    tm_o = orc.make_triad_mask(h, w, *c["triad"]) if c["triad"] else None
    vg_g = pc.make_vignette(h, w, c["vignette"]) if c["vignette"] else None
    vg_o = orc.make_vignette(h, w, c["vignette"]) if c["vignette"] else None
    kw = dict(noise_plane=plane) if c["noise_strength"] > 0 else {}
    ug, sg = pc.apply_crt_effect(*crt_args(frame, tm_g, vg_g, 0.0, None, 1.25, c), warp_strength=0.15, **kw)
    _, so = orc.apply_crt_effect(*crt_args(frame, tm_o, vg_o, 0.0, None, 1.25, c), warp_strength=0.15, **kw)
    assert ug.dtype == np.float16 and np.abs(sg.astype(np.float64) - so).max() <= 3e-7
    exp16 = np.abs(so.astype(np.float32) * np.float32(255.0)).astype(np.float16)
    assert np.abs(ug.astype(np.float32) - exp16.astype(np.float32)).max() <= 0.125      # one half ulp at 128..255
    assert (ug != exp16).mean() < 5e-3          # half is 32x finer than uint8 around 200: more last-bit flips per float ulp


@pytest.mark.parametrize("half", [False, True])
def test_banded_frames_equal_whole_frames(pc, half, monkeypatch):
    """A frame whose float32 pre-warp image does not fit the Infinity Cache (8K: 398 MB) runs as bands of row segments — k_phosphor over a
    band, then at once the k_warp_lean rows whose taps the band completes (crtfx_set_params plans the split from the barrel map).  BAND_MB = 1
    bands these small frames into 2 - 4 pieces: the frames must equal the whole-frame launches bit for bit, for barrel and pincushion maps,
    a strong warp (rows that need source rows far below them), both pixel formats, and frames too small to band."""
    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    dev = torch.device("cuda", torch.cuda.current_device())
    n = 3
    for (h, w, warp, sigma) in [(520, 448, 0.15, 3.0), (300, 704, -0.4, 1.2), (416, 512, 0.9, 2.0), (1000, 128, 0.15, 4.4), (64, 128, 0.15, 3.0)]:
        if half:
            frames = torch.from_numpy(np.stack([f16_frame(h, w, 40 + i) for i in range(n)])).to(dev)
        else:
            frames = torch.from_numpy(np.stack([make_frame(h, w, seed=40 + i, kind="grad") for i in range(n)])).to(dev)
        rs = RenderSettings(fast_bloom=False, bloom_sigma=sigma, pixel_size=1, persistence=0.0, warp_strength=warp)
        outs = {}
        for name, opts in (("whole", {"GROUP": 1, "BAND_MB": -1}), ("banded", {"GROUP": 1, "BAND_MB": 1}), ("banded_rows2", {"GROUP": 1, "BAND_MB": 1, "WARP_ROWS": 2}),
                           ("banded_seg", {"GROUP": 1, "BAND_MB": 1, "SEG_ROWS": 48}),
                           ("general_warp", {"GROUP": 1, "BAND_MB": -1, "NO_PLAIN_WARP": 1})):      # k_warp_lean's general build (per-lane 2-byte stores for half frames)
            monkeypatch.setattr(effects, "DEBUG_OPTIONS", dict(opts))
            effects._tls.engines = {}
            pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=11, dtype=frames.dtype)
            out, _ = pipe.run(frames, first_index=5)
            outs[name] = out.cpu().numpy()
        for name in ("banded", "banded_rows2", "banded_seg", "general_warp"):
            assert np.array_equal(outs["whole"].view(np.uint8), outs[name].view(np.uint8)), (h, w, warp, name)
    effects._tls.engines = {}


def test_frames_that_do_not_start_on_a_dword(pc):
    """A C-ABI caller may hand over frames at ANY byte address (include/crtfx.h asks for no alignment): k_phosphor_ct's A phase reads a
    frame row as aligned dwords, so a frame base or frame stride that is not a multiple of four takes the byte-wise k_phosphor_cc — and
    both give the bits an aligned copy of the same frames gives (the in / out buffers here start 1, 2 and 3 bytes into an allocation,
    and the odd-stride batch puts every second frame on another alignment)."""
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    from pythoncrt_amd import _lib
    dev = torch.device("cuda", torch.cuda.current_device())
    rs, _, _ = baseline_config(3)
    h, w, n = 72, 264, 3
    frames = torch.from_numpy(np.stack([make_frame(h, w, seed=700 + i, kind="grad") for i in range(n)])).to(dev)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=3)
    ref, _ = pipe.run(frames)
    nb = n * h * w * 3
    for shift in (1, 2, 3):
        raw_in = torch.zeros(nb + 8, dtype=torch.uint8, device=dev)
        raw_out = torch.zeros(nb + 8, dtype=torch.uint8, device=dev)
        fin = raw_in[shift:shift + nb].view(n, h, w, 3)
        fout = raw_out[shift:shift + nb].view(n, h, w, 3)
        fin.copy_(frames)
        assert fin.data_ptr() % 4 == shift
        got, _ = pipe.run(fin, out=fout)
        assert torch.equal(got, ref), shift
    # an odd frame stride through crtfx_process_batch itself: frames 5 bytes further apart than their size
    stride = h * w * 3 + 5
    raw_in = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    raw_out = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    for i in range(n):
        raw_in[i * stride:i * stride + h * w * 3].copy_(frames[i].reshape(-1))
    recs, hold = pipe.frame_records(0, n, None)
    recs = recs.ctypes.data_as(ctypes.POINTER(_lib.CrtfxFrame)) if isinstance(recs, np.ndarray) else recs
    rc = pipe.lib.crtfx_process_batch(pipe.engine.ctx, raw_in.data_ptr(), stride, raw_out.data_ptr(), stride, n, recs, None, 0.0, 0, None,
                                      torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0
    for i in range(n):
        assert torch.equal(raw_out[i * stride:i * stride + h * w * 3].view(h, w, 3), ref[i]), i


@pytest.mark.parametrize("persistence", [0.0, 0.5])
def test_grouped_batch_equals_frame_by_frame(pc, persistence, monkeypatch):
    """crtfx_process_batch launches up to 4 frames per grid (blockIdx.z = frame); the frames must come out
    exactly as when each is rendered by its own call, and in order under the persistence IIR."""
    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    monkeypatch.setattr(effects, "DEBUG_OPTIONS", {"GROUP": 4})
    effects._tls.engines = {}
    dev = torch.device("cuda", torch.cuda.current_device())
    rs, _, _ = baseline_config(2)
    rs.persistence = persistence
    h, w, n = 96, 200, 7
    frames = torch.from_numpy(np.stack([make_frame(h, w, seed=100 + i, kind="grad") for i in range(n)])).to(dev)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=9)
    out, state = pipe.run(frames)
    tm, vg = pc.make_triad_mask(h, w, rs.triad_strength, rs.triad_softness), pc.make_vignette(h, w, rs.vignette_strength)
    c = dict(BASE, scanline_strength=rs.scanline_strength, aberration_px=rs.aberration_px, bloom_sigma=rs.bloom_sigma,
             bloom_strength=rs.bloom_strength, noise_strength=rs.noise_strength)
    prev = None
    for i in range(n):
        st = pc.apply_static_effects(frames[i], c["scanline_strength"], tm, 2.2, False, c["aberration_px"], c["bloom_sigma"], c["bloom_strength"],
                                     0.0, c["noise_strength"], vg, 2.0, (i / 30.0) * rs.scanline_speed_px_s, False, 1, 0, 0.0,
                                     time_sec=i / 30.0, warp_strength=rs.warp_strength, noise_seed=9, frame_index=i)
        if prev is not None and persistence > 0.0:
            st = torch.clamp(np.float32(persistence) * prev + np.float32(1.0 - persistence) * st, 0.0, 1.0)
        prev = st
        u8 = torch.from_numpy(orc.convert_scale_abs(st.cpu().numpy())).to(dev)
        d = (out[i].to(torch.int16) - u8.to(torch.int16)).abs()
        if persistence == 0.0:
            assert torch.equal(out[i], u8), i
        else:
            assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-3, i     # blend runs in float64 on the GPU
    effects._tls.engines = {}


def test_fp16_normalise_exhaustive(pc):
    """Every finite half value / 255.0 (true float32 division) through the kernels' corrected-reciprocal form."""
    h16 = np.arange(65536, dtype=np.uint16).view(np.float16)
    h16 = h16[np.isfinite(h16.astype(np.float32))]
    n = h16.size                        # 63488 = 64 * 992
    frame = np.stack([h16, h16[::-1], np.roll(h16, 7)], axis=1).reshape(64, n // 64, 3)
    a = (frame, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0)
    got = pc.apply_static_effects(*a)
    exp = frame.astype(np.float32) / 255.0
    assert got.dtype == np.float32 and np.array_equal(got, exp)
    # and through the bloom kernels' staging / centre ring (sigma 3 -> lean half build is gate-specific, so this
    # takes the generic kernel; the lean half build is covered by test_fp16_frames cfg 1)
    small = frame[:, :130].copy()
    small = np.abs(small).astype(np.float16)
    small[small > 255] = 255
    g2, e2 = both_static(pc, small, dict(bloom_sigma=3.0, bloom_strength=0.25))
    assert_bit_exact(g2, e2)


# ---- preview path: the previous state arrives with another size (ref:689-690, cv2.resize INTER_LINEAR) ----------

@pytest.mark.parametrize("prev_hw", [(40, 56), (128, 192), (100, 75), (64, 200), (1, 1)])
@pytest.mark.parametrize("promoted", [False, True])
def test_state_prev_of_another_size(pc, prev_hw, promoted):
    """The window was resized between two ticks: state_prev is bilinearly resampled to the new frame size.  With an
    unpromoted chain the state is float32 on both sides and the result is bit-exact; with the vignette on the
    reference's state is float64 (held rounded to float32 on the GPU): 4e-7 / 1 LSB."""
    h, w = 64, 96
    ph, pw = prev_hw
    c = dict(BASE, scanline_strength=0.6, aberration_px=1, bloom_sigma=1.2, bloom_strength=0.25)
    vs = 0.25 if promoted else None
    mk = lambda mod, hh, ww: (mod.make_triad_mask(hh, ww, 0.35, 0.5), mod.make_vignette(hh, ww, vs) if vs else None)
    f0, f1 = make_frame(ph, pw, seed=81, kind="grad"), make_frame(h, w, seed=82, kind="grad")
    (tg0, vg0), (to0, vo0) = mk(pc, ph, pw), mk(orc, ph, pw)
    (tg1, vg1), (to1, vo1) = mk(pc, h, w), mk(orc, h, w)
    _, sg = pc.apply_crt_effect(*crt_args(f0, tg0, vg0, 0.5, None, 3.0, c))
    _, so = orc.apply_crt_effect(*crt_args(f0, to0, vo0, 0.5, None, 3.0, c))
    assert sg.shape == (ph, pw, 3) and np.array_equal(sg, so.astype(np.float32))
    ug, sg1 = pc.apply_crt_effect(*crt_args(f1, tg1, vg1, 0.5, sg, 4.0, c))
    uo, so1 = orc.apply_crt_effect(*crt_args(f1, to1, vo1, 0.5, so, 4.0, c))
    assert sg1.shape == (h, w, 3) and sg1.dtype == np.float32
    if not promoted:
        assert so1.dtype == np.float32
        assert np.array_equal(sg1, so1) and np.array_equal(ug, uo)
    else:
        assert so1.dtype == np.float64
        assert np.abs(sg1.astype(np.float64) - so1).max() <= 4e-7
        d = np.abs(ug.astype(np.int16) - uo.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
    # the resize entry point alone, against the oracle's restatement of cv2.resize on the same float32 state
    dev = torch.device("cuda", 0)
    from pythoncrt_amd.effects import _engine
    eng = _engine(dev, h, w)
    dst = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    sg = np.asarray(sg)                  # the numpy path hands back a DeviceState: materialise it
    src = torch.from_numpy(sg).to(dev)
    rc = eng.lib.crtfx_resize_state(eng.ctx, src.data_ptr(), ph, pw, dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    exp = orc.resize(sg.astype(np.float64) if promoted else sg, (w, h), "linear")
    assert np.array_equal(dst.cpu().numpy(), exp.astype(np.float32))


# ---- the reference's frame-parallel dispatcher shape (a16, ref:1015-1131) over the drop-in -------------------

def test_two_worker_threads_like_process_video(pc):
    """process_video calls apply_static_effects from a 2-thread pool (17 positional arguments, the rest by keyword,
    ref:1045-1078) with shared read-only masks, keeps <= 4*workers futures in flight and commits strictly in
    order.  The drop-in keeps one ctx per (device, size, thread): results must not depend on which thread ran a
    frame, and the in-order commit must reproduce the single-threaded render."""
    from concurrent.futures import ThreadPoolExecutor
    h, w, n = 90, 160, 12
    frames = [make_frame(h, w, seed=90 + i, kind="grad") for i in range(n)]
    tm, vg = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.25)
    fps, speed, p = 30.0, 30.0, 0.4

    def job(i, f):
        return pc.apply_static_effects(f, 0.6, tm, 2.2, False, 1, 1.2, 0.25, 0.0, 1.5, vg, 2.0, (i / fps) * speed, False, 1, 0, 0.0,
                                       time_sec=i / fps, flicker_strength=0.3, flicker_hz=7.0, warp_strength=0.15,
                                       noise_seed=5, frame_index=i)

    def commit(results):
        prev, outs = None, []
        for i in range(n):
            prev, u8 = orc.persistence_blend(prev, results[i], p)      # ref:1086-1098 (numpy blend + convertScaleAbs)
            outs.append(u8)
        return np.stack(outs)

    single = {i: job(i, frames[i]) for i in range(n)}
    workers, queue_cap = 2, 8
    futures, got, nxt = {}, {}, 0
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for i, f in enumerate(frames):
            futures[i] = ex.submit(job, i, f)
            while len(futures) >= queue_cap or nxt in futures:
                if nxt in futures:
                    got[nxt] = futures.pop(nxt).result()
                    nxt += 1
                else:
                    break
        while nxt in futures:
            got[nxt] = futures.pop(nxt).result()
            nxt += 1
    assert sorted(got) == list(range(n))
    for i in range(n):
        assert got[i].dtype == np.float32 and np.array_equal(got[i], single[i]), i
    assert np.array_equal(commit(got), commit(single))


def test_scanline_plane_on_device(pc):
    """make_scanline_mask_2d (ref:308-328) generated by crtfx_scanline_plane against the oracle's table (the
    reference's numpy expression): float64 sin/pow of the device are not numpy's, so single values may land one
    float32 ulp away; the bar is <= 1 ulp anywhere and <= 1e-5 of the elements off at all."""
    from pythoncrt_amd import tables
    from pythoncrt_amd.effects import _engine
    dev = torch.device("cuda", torch.cuda.current_device())
    for (h, w, strength, period, phase, angle, thick) in [(270, 480, 0.6, 2.0, 3.25, 12.0, 1.0), (135, 333, 1.0, 3.7, -17.5, -40.0, 2.5),
                                                          (64, 64, 0.35, 1.0, 250.0, 0.0, 0.3), (1080, 1920, 0.6, 2.0, 41.0, 7.0, 4.0)]:
        eng = _engine(dev, h, w)
        out = torch.empty((h, w), dtype=torch.float32, device=dev)
        omega, tan_t, inv_sharp = tables.scanline_plane_scalars(period, angle, thick)
        rc = eng.lib.crtfx_scanline_plane(eng.ctx, strength, omega, phase, tan_t, inv_sharp, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        got = out.cpu().numpy()
        exp = orc.make_scanline_mask_2d(h, w, strength, period, phase, angle, thick)
        off = got != exp
        assert off.mean() <= 1e-5, off.mean()
        if off.any():
            ulp = np.spacing(np.maximum(np.abs(exp[off]), np.float32(2.0 ** -20)))
            assert (np.abs(got[off].astype(np.float64) - exp[off]) <= ulp).all()


# ---- BASELINE full sizes through the render loop ----------------------------------------------------------------

def _export_planes(pipe, seed, first, n, h, w):
    out = []
    for i in range(n):
        p = torch.empty((h, w), dtype=torch.float32, device=pipe.device)
        assert pipe.lib.crtfx_noise_plane(pipe.engine.ctx, seed, first + i, p.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        out.append(p.cpu().numpy())
    return out


def test_4k_render_loop_against_oracle(pc):
    """BASELINE configs[2] (the bench workload: 4K, bloom sigma 3, warp 0.15) through crtfx_process_batch — the grouped
    launch, k_phosphor_ct<9>, the branch-free k_warp_lean, the in-kernel grain — against the oracle's in-order render
    of the same two frames (about ten seconds of CPU)."""
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    rs, h, w = baseline_config(3)
    dev = torch.device("cuda", torch.cuda.current_device())
    n, first, seed = 2, 7, 4242
    frames = np.stack([make_frame(h, w, seed=300 + i, kind="grad") for i in range(n)])
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=seed)
    out, _ = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    planes = _export_planes(pipe, seed, first, n, h, w)
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma",
                                          "bloom_strength", "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom",
                                          "pixel_size", "warp_strength")}
    exp, _ = orc.process_frames(list(frames), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                rs.vignette_strength, noise_planes=planes, first_index=first)
    d = np.abs(out.cpu().numpy().astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3, (int(d.max()), float((d != 0).mean()))


@pytest.mark.parametrize("sigma,warp", [(11.0, 0.15), (45.0, 0.15), (3.0, 0.0), (1.2, 0.0), (11.0, 0.0)])          # radii 33, 135 | 9, 4, 33 with the commit-only k_warp_lean
def test_render_loop_any_sigma(pc, sigma, warp):
    """The render loop (crtfx_process_batch: warp, persistence 0.5, in-kernel grain) with a bloom sigma beyond the GUI's
    range — the split path behind the same entry point — against the oracle's in-order render (ref:609-610 takes any sigma);
    and with the warp off: the persistence chain behind the Gaussian kernels then runs in the commit-only build of k_warp_lean."""
    import dataclasses
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    rs = dataclasses.replace(baseline_config(4)[0], bloom_sigma=sigma, warp_strength=warp)
    h, w = 120, 200
    dev = torch.device("cuda", torch.cuda.current_device())
    n, first, seed = 3, 5, 99
    frames = np.stack([make_frame(h, w, seed=400 + i, kind="grad") for i in range(n)])
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=seed)
    out, _ = pipe.run(torch.from_numpy(frames).to(dev), first_index=first)
    planes = _export_planes(pipe, seed, first, n, h, w)
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma",
                                          "bloom_strength", "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom",
                                          "pixel_size", "warp_strength")}
    exp, _ = orc.process_frames(list(frames), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                                rs.vignette_strength, noise_planes=planes, first_index=first)
    d = np.abs(out.cpu().numpy().astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3, (int(d.max()), float((d != 0).mean()))


def test_8k_fp16_batch_properties(pc):
    """BASELINE configs[4]: 8K frames held as float16 in and out.  A batch equals its frames run one by one, is
    deterministic, finite, and its top-left 64 x 64 block (where the warp samples outside the frame) is zero."""
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    rs, h, w = baseline_config(5)
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(8)
    frames = (torch.rand((2, h, w, 3), generator=g) * 255.0).to(torch.float16).to(dev)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=77, dtype=torch.float16)
    both, _ = pipe.run(frames, first_index=3)
    again, _ = pipe.run(frames, first_index=3)
    assert both.dtype == torch.float16 and torch.equal(both, again)
    for i in range(2):
        one, _ = pipe.run(frames[i:i + 1], first_index=3 + i)
        assert torch.equal(one[0], both[i])
    assert bool(torch.isfinite(both.float()).all()) and float(both.min()) >= 0.0 and float(both.max()) <= 255.0
    assert not bool(both[:, :64, :64].any())
    assert float(both[:, h // 2 - 200:h // 2 + 200, w // 2 - 200:w // 2 + 200].float().mean()) > 20.0


def test_engine_cache_is_bounded(pc):
    """A preview window dragged through many sizes: the per-thread ctx cache keeps the most recent few and frees the rest."""
    from pythoncrt_amd import effects
    a = lambda f: (f, 0.6, None, 2.2, False, 1, 1.2, 0.25, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0)
    first = make_frame(20, 30, seed=1)
    ref = pc.apply_static_effects(*a(first))
    for k in range(10):
        pc.apply_static_effects(*a(make_frame(21 + k, 40 + 3 * k, seed=2 + k)))
    assert len(effects._tls.engines) <= effects._ENGINES_PER_THREAD
    assert (torch.cuda.current_device(), 20, 30, 0) not in effects._tls.engines       # evicted ...
    assert np.array_equal(pc.apply_static_effects(*a(first)), ref)                    # ... and rebuilt on demand


@pytest.mark.parametrize("grade", [dict(gamma=1.8), dict(gamma=0.6, brightness=0.04, contrast=1.15), dict(temperature=0.6, gamma=2.2),
                                   dict(temperature=-0.8, brightness=-0.05, contrast=0.9)])
@pytest.mark.parametrize("sigma", [0.0, 1.2, 3.0])
def test_grade_table_paths(pc, grade, sigma):
    """uint8 frames with the saturation mix off: a1 + a4 come from the 3 x 256 table built with the reference's numpy
    expressions (register-window kernel: LDS copy; pointwise kernels with --gamma: L1 reads).  Bit-exact without gamma;
    with gamma the table holds numpy's own powf values, compared at the usual 3e-7."""
    frame = make_frame(70, 130, seed=120)
    cfg = dict(aberration_px=1, bloom_sigma=sigma, bloom_strength=0.25 if sigma > 0 else 0.0, scanline_strength=0.6, scanline_phase_px=2.0)
    got, exp = run_both(pc, frame, cfg, **grade)
    if "gamma" in grade:
        assert np.abs(got - exp.astype(np.float32)).max() <= 3e-7
    else:
        assert_bit_exact(got, exp)


def test_reference_built_mask_arrays_take_the_fast_path(pc):
    """A caller that rebinds only apply_* hands over the reference's own mask arrays (H x W x 3 float32 with identical
    rows; H x W float64 = 1 - s * clip(nx^2 + ny^2)).  They are recognised (verified element for element) and use the
    row / analytic path: same bits as with the descriptors, no per-pixel planes in the parameter block.  A mask that
    does not have that structure stays a plane."""
    from pythoncrt_amd import effects
    h, w = 90, 160
    frame = make_frame(h, w, seed=130, kind="grad")
    tm_d, vg_d = pc.make_triad_mask(h, w, 0.35, 0.5), pc.make_vignette(h, w, 0.35)
    tm_a, vg_a = orc.make_triad_mask(h, w, 0.35, 0.5), orc.make_vignette(h, w, 0.35)      # what the reference's builders return
    assert isinstance(tm_a, np.ndarray) and tm_a.shape == (h, w, 3) and vg_a.dtype == np.float64
    a = lambda tm, vg: (frame, 0.6, tm, 2.2, False, 1, 1.2, 0.25, 0.0, 1.5, vg, 2.0, 1.0, False, 1, 0, 0.0)
    kw = dict(noise_seed=3, frame_index=1, warp_strength=0.15)
    ref = pc.apply_static_effects(*a(tm_d, vg_d), **kw)
    got = pc.apply_static_effects(*a(tm_a, vg_a), **kw)
    assert np.array_equal(got, ref)
    eng = effects._engine(torch.device("cuda", torch.cuda.current_device()), h, w)
    assert "triad_row" in eng.keep and "triad_full" not in eng.keep and "nx2" in eng.keep and "vig_full" not in eng.keep
    assert isinstance(effects._recognise_vignette(vg_a), effects.VignetteMask) and effects._recognise_vignette(vg_a).strength == 0.35
    # an arbitrary mask is left alone and still works (per-pixel plane path), against the oracle
    rng = np.random.default_rng(131)
    tm_x = (0.5 + 0.5 * rng.random((h, w, 3))).astype(np.float32)
    vg_x = 0.5 + 0.5 * rng.random((h, w))
    assert effects._recognise_triad(tm_x) is tm_x and effects._recognise_vignette(vg_x) is vg_x
    plane = rng.standard_normal((h, w), dtype=np.float32)
    g = pc.apply_static_effects(*a(tm_x, vg_x), noise_plane=plane)
    o = orc.apply_static_effects(*a(tm_x, vg_x), noise_plane=plane)
    assert np.array_equal(g, o.astype(np.float32))


def test_baseline_config_1_720p(pc):
    """BASELINE configs[0]: 1280 x 720, everything off except scanlines 0.6 / period 2 / speed 30 at 30 fps (phase = i):
    the reference's CPU-runnable case.  Scanline rows come from the reference's numpy expression, so the frames are
    the oracle's bit for bit."""
    from pythoncrt_amd.pipeline import FramePipeline, baseline_config
    rs, h, w = baseline_config(1)
    assert (h, w) == (720, 1280)
    g = torch.Generator(device="cpu").manual_seed(1234)
    frames = torch.randint(0, 256, (5, h, w, 3), dtype=torch.uint8, generator=g)
    dev = torch.device("cuda", torch.cuda.current_device())
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=0)
    out, _ = pipe.run(frames.to(dev), first_index=0)
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma",
                                          "bloom_strength", "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size")}
    exp, _ = orc.process_frames(list(frames.numpy()), params, 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength,
                                rs.triad_softness, rs.vignette_strength)
    assert np.array_equal(out.cpu().numpy(), np.stack(exp))
