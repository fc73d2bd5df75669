"""The oracle (oracle/crt_oracle.py) against outputs of the reference's own function bodies
(tests/golden/reference_numpy_stages.npz, made by tests/golden/gen_golden.py).

Integer/index work is compared bit-exactly.  Stages that go through np.sin / np.power /
np.tan are compared to <= 2 float ulp: numpy dispatches those to different SIMD kernels on
different host CPUs, so the fixture (made in the build container) and a run on another
machine may legitimately differ in the last bit."""
import numpy as np
import pytest

from oracle import crt_oracle as orc


def ulp_close(a, b, ulps=2):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert a.shape == b.shape
    eps = np.finfo(a.dtype).eps
    tol = ulps * eps * np.maximum(np.abs(a), np.abs(b)) + ulps * np.finfo(a.dtype).tiny
    bad = np.abs(a.astype(np.float64) - b.astype(np.float64)) > tol
    assert not bad.any(), f"{bad.sum()} of {bad.size} beyond {ulps} ulp; max abs diff {np.abs(a - b).max()}"


def frames(h, w):
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = np.stack([(xx * 255) // max(1, w - 1), (yy * 255) // max(1, h - 1), ((xx + yy) * 255) // max(1, h + w - 2)], axis=2).astype(np.uint8)
    return {"noise": noise, "grad": grad}


def test_shift_channel_bit_exact(golden):
    plane = golden["shift/in"]
    for dx in (-8, -1, 1, 3, 8):
        got = orc.shift_channel(plane, dx, 0)
        assert np.array_equal(got, golden[f"shift/dx{dx}"])
        # explicit index statement of the roll: out[y, x] = in[y, (x - dx) mod W]
        w = plane.shape[1]
        assert np.array_equal(got, plane[:, (np.arange(w) - dx) % w])


def test_scanline_1d(golden):
    for i, (h, s, p, ph) in enumerate(golden["scan1d/args"]):
        got = orc.make_scanline_mask_dynamic(int(h), float(s), float(p), float(ph))  # python floats, as the reference callers pass
        assert got.dtype == np.float32
        ulp_close(got, golden[f"scan1d/{i}"])


def test_scanline_2d(golden):
    for i, a in enumerate(golden["scan2d/args"]):
        got = orc.make_scanline_mask_2d(int(a[0]), int(a[1]), *map(float, a[2:]))
        ulp_close(got, golden[f"scan2d/{i}"])


def test_triad_mask_and_ksize(golden):
    assert np.array_equal(orc.make_triad_mask(5, 20, 0.35, 0.0), golden["triad_mask/s0"])
    for s, k in zip(golden["triad_mask/soft"], golden["triad_mask/soft_ksize"]):
        assert (orc.triad_ksize(float(s)), 1) == tuple(k)
    # analytic KAT: strength 0.35 row = [1,.65,.65],[.65,1,.65],[.65,.65,1]
    row = orc.make_triad_row(6, 0.35)[0]
    assert np.array_equal(row[:3], np.array([[1, .65, .65], [.65, 1, .65], [.65, .65, 1]], np.float32))


def test_vignette_bit_exact(golden):
    for i, (h, w, s) in enumerate(golden["vignette/args"]):
        got = orc.make_vignette(int(h), int(w), float(s))
        assert got.dtype == np.float64
        assert np.array_equal(got, golden[f"vignette/{i}"])
    assert orc.make_vignette(48, 64, 0.25)[0, 0] == 0.75  # corner KAT: r2 clipped to 1


def test_colour_grade(golden):
    img0 = golden["grade/in_u8"].astype(np.float32) / 255.0
    for i, (b, c, g, s, t) in enumerate(golden["grade/args"]):
        got = orc.apply_color_adjustments(img0.copy(), float(b), float(c), float(g), float(s), float(t))
        exp = golden[f"grade/{i}"]
        if g == 1.0:
            assert np.array_equal(got, exp), i
        else:
            ulp_close(got, exp)


def test_triad_apply(golden):
    img0 = golden["grade/in_u8"].astype(np.float32) / 255.0
    mask = orc.make_triad_mask(48, 64, 0.35, 0.0)
    for i, (g, p) in enumerate(golden["triad_apply/args"]):
        got = orc.apply_triad_mask(img0.copy(), mask, float(g), bool(p))
        # LUT entries come from np.power: a 1-ulp LUT difference moves outputs by 1 ulp
        ulp_close(got, golden[f"triad_apply/{i}"])
    # LUT index KAT
    x = np.array([0, .5, .999, 1, 127 / 255], np.float32)
    idx = np.clip((np.clip(x, 0, 1) * 1024.0).astype(np.int32), 0, 1024)
    assert idx.tolist() == [0, 512, 1022, 1024, 509]


def test_barrel_maps_bit_exact(golden):
    for i, (h, w, s) in enumerate(golden["warpmap/args"]):
        mx, my = orc.barrel_maps(int(h), int(w), float(s))
        assert mx.dtype == np.float32
        assert np.array_equal(mx, golden[f"warpmap/{i}/x"])
        assert np.array_equal(my, golden[f"warpmap/{i}/y"])


def test_bloom_ksize_and_threshold_src(golden):
    for s, k in zip(golden["bloom/sigma"], golden["bloom/ksize"]):
        assert (orc.bloom_ksize(float(s)),) * 2 == tuple(k)
    assert orc.bloom_ksize(3.0) == 19 and orc.bloom_ksize(1.2) == 9 and orc.bloom_ksize(1.5) == 9
    # thresholded blur source: reproduce the statements up to the GaussianBlur call
    fr = frames(48, 64)["noise"]
    img = fr.astype(np.float32) / 255.0
    img = np.stack([orc.shift_channel(img[:, :, 0], 1, 0), img[:, :, 1], orc.shift_channel(img[:, :, 2], -1, 0)], axis=2)
    img = orc.apply_color_adjustments(img, 0.05, 1.1, 1.0, 1.0, 0.0)
    thr = 0.4
    src = np.clip((img - thr) / max(1e-6, 1.0 - thr), 0.0, 1.0)
    assert np.array_equal(src, golden["bloom/src_thr0.4"])


CHAINS = {
    "scan_only_p0": dict(scanline_strength=0.6, triad=None, vig=None, aberration_px=0, scanline_phase_px=0.0),
    "scan_only_p7": dict(scanline_strength=0.6, triad=None, vig=None, aberration_px=0, scanline_phase_px=7.0),
    "numpy_full": dict(scanline_strength=0.6, triad=(0.35, 0.0), vig=0.25, aberration_px=1, scanline_phase_px=1.25,
                       triad_gamma=2.2, triad_preserve_luma=False, time_sec=0.3, flicker_strength=0.5, flicker_hz=7.0),
    "numpy_full_luma": dict(scanline_strength=0.6, triad=(0.35, 0.0), vig=0.25, aberration_px=-3, scanline_phase_px=4.5,
                            triad_gamma=2.2, triad_preserve_luma=True, brightness=0.03, contrast=1.1, gamma=1.4,
                            saturation=1.2, temperature=0.3),
    "scan2d_vig": dict(scanline_strength=0.7, triad=None, vig=0.6, aberration_px=2, scanline_phase_px=3.0,
                       scanline_angle=7.5, scanline_thickness=1.8),
    "glitch_render": dict(scanline_strength=0.6, triad=(0.5, 0.0), vig=None, aberration_px=1, scanline_phase_px=13.0,
                          glitch_amp_px=9, glitch_height_frac=0.4),
}


def run_chain(frame, c):
    h, w = frame.shape[:2]
    tm = orc.make_triad_mask(h, w, *c["triad"]) if c.get("triad") else None
    vg = orc.make_vignette(h, w, c["vig"]) if c.get("vig") else None
    return orc.apply_static_effects(
        frame, c["scanline_strength"], tm, c.get("triad_gamma", 2.2), c.get("triad_preserve_luma", False),
        c["aberration_px"], 0.0, 0.0, 0.0, 0.0, vg, 2.0, c["scanline_phase_px"], False, 1,
        c.get("glitch_amp_px", 0), c.get("glitch_height_frac", 0.0), time_sec=c.get("time_sec", 0.0),
        brightness=c.get("brightness", 0.0), contrast=c.get("contrast", 1.0), gamma=c.get("gamma", 1.0),
        saturation=c.get("saturation", 1.0), temperature=c.get("temperature", 0.0),
        flicker_strength=c.get("flicker_strength", 0.0), flicker_hz=c.get("flicker_hz", 0.0),
        scanline_angle=c.get("scanline_angle", 0.0), scanline_thickness=c.get("scanline_thickness", 1.0))


@pytest.mark.parametrize("cname", sorted(CHAINS))
def test_chain_against_reference(golden, cname):
    keys = [k for k in golden.files if k.startswith("chain/") and k.endswith("/" + cname)]
    assert keys
    for k in keys:
        _, size, fname, _ = k.split("/")
        h, w = map(int, size.split("x"))
        got = run_chain(frames(h, w)[fname], CHAINS[cname])
        exp = golden[k]
        assert got.dtype == exp.dtype, (k, got.dtype, exp.dtype)
        # values are bounded by 1: compare absolutely at 2 float32 ulp of 1.0 (sin/pow tables)
        assert np.abs(got.astype(np.float64) - exp.astype(np.float64)).max() <= 2.4e-7, k


def test_chain_dtype_promotion(golden):
    """float32 until the f64 vignette / np.float64 flicker factor, float64 after (NumPy 2)."""
    f = frames(48, 64)["noise"]
    assert run_chain(f, CHAINS["scan_only_p0"]).dtype == np.float32
    assert run_chain(f, CHAINS["numpy_full"]).dtype == np.float64
    c = dict(CHAINS["scan_only_p0"], time_sec=0.1, flicker_strength=0.5, flicker_hz=3.0)
    assert run_chain(f, c).dtype == np.float64


def test_preview_glitch_float(golden):
    f = frames(48, 64)["noise"]
    out, state = orc.apply_crt_effect(f, 0.6, None, 2.2, False, 1, 0.0, 0.0, 0.0, 0.0, None, 0.0, None, 2.0, 250.0,
                                      False, 1, glitch_amp_px=11, glitch_height_frac=0.5)
    exp = golden["chain/48x64/noise/glitch_preview_float"]
    assert state.dtype == exp.dtype
    assert np.abs(state - exp).max() <= 2.4e-7
    assert out.dtype == np.uint8 and out.shape == f.shape
