"""The oracle's restatements of the OpenCV calls on the hot path (parity UNPINNED against cv2
itself: it is not installable here and the reference holds no vectors) cross-checked against
independent implementations (scipy.ndimage, torch CPU) and analytic known answers."""
import numpy as np
import scipy.ndimage as ndi
import torch
import torch.nn.functional as F

from oracle import crt_oracle as orc


def test_gaussian_kernel_properties():
    for k, s in [(19, 3.0), (9, 1.2), (9, 1.5), (5, 0.5), (3, 0.2), (25, 4.0), (61, 10.0)]:
        t = orc.gaussian_kernel(k, s)
        assert t.dtype == np.float32 and t.shape == (k,)
        assert np.array_equal(t, t[::-1])
        x = np.arange(k) - (k - 1) / 2
        ref = np.exp(-x * x / (2 * s * s))
        ref /= ref.sum()
        assert np.abs(t - ref).max() < 1e-7
        assert abs(float(t.astype(np.float64).sum()) - 1.0) < 1e-6


def test_blur_impulse_is_outer_product_of_taps():
    k, s = 19, 3.0
    img = np.zeros((41, 45, 3), np.float32)
    img[20, 22] = 1.0
    out = orc.gaussian_blur(img, (k, k), s, s)
    t = orc.gaussian_kernel(k, s)
    exp = np.outer(t, t)
    for c in range(3):
        assert np.array_equal(out[20 - 9:20 + 10, 22 - 9:22 + 10, c], exp.astype(np.float32))
    assert out[:11].sum() == 0 and out[:, :13].sum() == 0


def test_blur_matches_scipy_and_numpy_twin():
    rng = np.random.default_rng(3)
    img = rng.random((37, 53, 3), dtype=np.float32)
    for k, s in [(19, 3.0), (9, 1.2), (5, 0.5), (1, 0.1)]:
        out = orc.gaussian_blur(img, (k, k), s, s)
        if k == 1:
            assert np.array_equal(out, img)
            continue
        t = orc.gaussian_kernel(k, s).astype(np.float64)
        ref = ndi.correlate1d(ndi.correlate1d(img.astype(np.float64), t, axis=1, mode="nearest"), t, axis=0, mode="nearest")
        assert np.abs(out - ref).max() < 5e-7
        twin = orc.gaussian_blur_slow(img, orc.gaussian_kernel(k, s), orc.gaussian_kernel(k, s))
        assert np.abs(out - twin).max() <= 6e-8  # fma vs double-rounded emulation: at most a tie


def test_blur_horizontal_only_is_row_independent():
    """ksize (k,1) as at ref:234: every row filtered alone, the column kernel is [1]."""
    rng = np.random.default_rng(4)
    img = rng.random((6, 31, 3), dtype=np.float32)
    out = orc.gaussian_blur(img, (5, 1), 0.5, 0.0)
    for y in range(6):
        assert np.array_equal(out[y:y + 1], orc.gaussian_blur(img[y:y + 1], (5, 1), 0.5, 0.0))
    m = orc.make_triad_mask(7, 30, 0.35, 0.5)
    assert all(np.array_equal(m[0], m[y]) for y in range(7))


def test_remap_identity_border_and_scipy():
    rng = np.random.default_rng(5)
    h, w = 23, 31
    img = rng.random((h, w, 3), dtype=np.float32)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    assert np.array_equal(orc.remap_bilinear(img, xs, ys), img)
    # far outside: all four taps out of range -> border value 0
    assert not orc.remap_bilinear(img, xs + 1000, ys).any()
    # half-pixel shift: exact average of horizontal neighbours, last column blends with 0
    out = orc.remap_bilinear(img, xs + 0.5, ys)
    assert np.array_equal(out[:, :-1], img[:, :-1] * np.float32(0.5) + img[:, 1:] * np.float32(0.5))
    assert np.array_equal(out[:, -1], img[:, -1] * np.float32(0.5))
    # general map against scipy on the 1/32-px quantised coordinates
    mx, my = orc.barrel_maps(h, w, 0.6)
    ix, iy, fxy = orc.remap_quantise(mx, my)
    qx = ix + (fxy & 31) / 32.0
    qy = iy + (fxy >> 5) / 32.0
    out = orc.remap_bilinear(img, mx, my)
    for c in range(3):
        ref = ndi.map_coordinates(img[:, :, c].astype(np.float64), [qy, qx], order=1, mode="grid-constant", cval=0.0)
        assert np.abs(out[:, :, c] - ref).max() < 3e-7
    out64 = orc.remap_bilinear(img.astype(np.float64), mx, my)
    assert out64.dtype == np.float64 and np.abs(out64 - out).max() < 3e-7


def test_remap_quantise_ties_to_even():
    mx = np.array([[0.015625, 0.046875, -0.015625, 5.0, 2.984375]], np.float32)  # *32 = .5, 1.5, -.5, 160, 95.5
    ix, iy, fxy = orc.remap_quantise(mx, np.zeros_like(mx))
    sx = ix * 32 + (fxy & 31)
    assert sx.tolist() == [[0, 2, 0, 160, 96]]


def test_resize_nearest_and_linear():
    rng = np.random.default_rng(6)
    img = rng.random((24, 36, 3), dtype=np.float32)
    # pixelate pair (ref:582-583): down by floor, up by floor
    small = orc.resize(img, (36 // 3, 24 // 3), "nearest")
    assert np.array_equal(small, img[::3, ::3])
    big = orc.resize(small, (36, 24), "nearest")
    assert np.array_equal(big, small[np.arange(24) // 3][:, np.arange(36) // 3])
    # fast-bloom pair (ref:606-607): exact 2x decimation = 2x2 mean; 2x upsample = half-pixel bilinear
    ds = orc.resize(img, (18, 12), "linear")
    ref = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) * np.float32(0.25)
    assert np.array_equal(ds, ref)
    up = orc.resize(ds, (36, 24), "linear")
    t = torch.from_numpy(ds).permute(2, 0, 1)[None]
    ref = F.interpolate(t, size=(24, 36), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(up - ref).max() < 3e-7
    # odd source: w//2 is not an exact 2x ratio -> generic bilinear path
    odd = rng.random((9, 11), dtype=np.float32)
    ds = orc.resize(odd, (5, 4), "linear")
    t = torch.from_numpy(odd)[None, None]
    ref = F.interpolate(t, size=(4, 5), mode="bilinear", align_corners=False)[0, 0].numpy()
    assert np.abs(ds - ref).max() < 1e-6  # the sample coordinate itself is a float32 here


def test_convert_scale_abs_rounding():
    x = np.array([0.5, 1.5, 2.5, 254.5, 255.5, 300.0, -3.5, -0.4], np.float32)
    assert orc.convert_scale_abs(x, 1.0).tolist() == [0, 2, 2, 254, 255, 255, 4, 0]
    assert orc.convert_scale_abs(x.astype(np.float64), 1.0).tolist() == [0, 2, 2, 254, 255, 255, 4, 0]
    u = np.arange(256, dtype=np.uint8)
    assert np.array_equal(orc.convert_scale_abs(u.astype(np.float32) / 255.0), u)  # u8 -> float -> u8 round trip


def test_add_weighted_and_persistence_blend():
    rng = np.random.default_rng(7)
    a = rng.random((5, 7, 3), dtype=np.float32)
    b = rng.random((5, 7, 3), dtype=np.float32)
    out = orc.add_weighted(a, 0.3, b, 0.7)
    assert out.dtype == np.float32 and np.abs(out - (0.3 * a + 0.7 * b)).max() < 1.2e-7
    st, u8 = orc.persistence_blend(None, a, 0.5)
    assert st is a and np.array_equal(u8, orc.convert_scale_abs(a))
    st2, _ = orc.persistence_blend(a, b, 0.5)
    assert np.array_equal(st2, np.clip(0.5 * a + 0.5 * b, 0, 1))


def test_full_chain_runs_with_cv_stages():
    """BASELINE config-2-shaped parameters at a toy size: exercises bloom, softened triad, grain
    plane injection and the warp through the restated OpenCV ops."""
    rng = np.random.default_rng(8)
    h, w = 40, 56
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    tm = orc.make_triad_mask(h, w, 0.35, 0.5)
    vg = orc.make_vignette(h, w, 0.25)
    noise = rng.standard_normal((h, w), dtype=np.float32)
    img = orc.apply_static_effects(frame, 0.6, tm, 2.2, False, 1, 1.2, 0.25, 0.0, 1.5, vg, 2.0, 1.25, False, 1, 0, 0.0,
                                   warp_strength=0.15, noise_plane=noise)
    assert img.dtype == np.float64 and img.shape == (h, w, 3)
    assert img.min() >= 0.0 and img.max() <= 1.0
    assert not img[0, 0].any()  # barrel warp 0.15 pulls the corners from outside the frame
    out, state = orc.apply_crt_effect(frame, 0.6, tm, 2.2, False, 1, 1.2, 0.25, 0.0, 1.5, vg, 0.5, img, 2.0, 1.25, False, 1,
                                      warp_strength=0.15, noise_plane=noise)
    assert out.dtype == np.uint8 and np.abs(state - img).max() < 1e-12  # same frame blended with itself
