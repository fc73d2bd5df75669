"""ORACLE — test infrastructure only, never the product.

CPU restatement (numpy + oracle/crt_oracle.c) of the per-frame CRT effect chain of
jaylikesbunda/PythonCRT (`crt_filter.py`, cited as `ref:LINE`).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module, and
only as the checker / the timed CPU baseline.  `pythoncrt_amd/` must never import it.

Parity status
-------------
* numpy-only stages (normalise, aberration, colour grade, triad mask s=0, triad LUT apply,
  1-D/2-D scanline masks, vignette, flicker, render-path persistence, both glitch variants and
  the chain driver `apply_static_effects` on cv2-free parameter sets) are PINNED: they are
  checked against outputs of the reference's own function bodies (tests/golden/*.npz, made by
  tests/golden/gen_golden.py, which loads the function defs from the reference source text by
  AST and never executes the module top level — ref:47 would shell out to pip).
* OpenCV-backed stages — cv2.GaussianBlur (ref:234,610,780), cv2.remap (ref:347), cv2.resize
  (ref:582-583,606-607,642,690,751-752,776-777,812), cv2.randn (ref:641,645,811,815),
  cv2.addWeighted (ref:693), cv2.convertScaleAbs (ref:696,1098,1124) — are PARITY UNPINNED:
  the arithmetic lives in opencv-python-headless>=4.8.0 (requirements.txt:6, not vendored, not
  installable here; the reference ships no tests or golden vectors).  They restate OpenCV 4.x's
  published algorithms (see crt_oracle.c header) and are cross-checked against scipy/torch.
* cv2.randn is not reproducible even in the reference (unseeded thread-local RNG), so the
  oracle takes the N(0,1) plane as an argument (`noise_plane`).

dtype flow follows NumPy 2 promotion (numpy 2.2 is what runs here): python floats are weak, the
f64 vignette mask (ref:267-275) and the np.float64 flicker factor (ref:632) promote the image to
float64 for every later stage.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """Load (building on first use) oracle/libcrt_oracle.so."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libcrt_oracle.so")
    src = os.path.join(_HERE, "crt_oracle.c")
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", _HERE, "libcrt_oracle.so"], check=True)
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int32)
    up = ctypes.POINTER(ctypes.c_uint8)
    ci = ctypes.c_int
    lib.orc_sepblur_f32.argtypes = [fp, fp, fp, ci, ci, ci, fp, ci, fp, ci]
    lib.orc_remap_quantise.argtypes = [fp, fp, ci, ip, ip, ip]
    lib.orc_remap_bilinear_f32.argtypes = [fp, fp, ci, ci, ci, fp, fp]
    lib.orc_remap_bilinear_f64.argtypes = [dp, dp, ci, ci, ci, fp, fp]
    lib.orc_resize_nearest_f32.argtypes = [fp, ci, ci, fp, ci, ci, ci]
    lib.orc_resize_linear_f32.argtypes = [fp, ci, ci, fp, ci, ci, ci]
    lib.orc_resize_linear_f32_variant.argtypes = [fp, ci, ci, fp, ci, ci, ci, ci]
    lib.orc_resize_linear_f64.argtypes = [dp, ci, ci, dp, ci, ci, ci]
    lib.orc_convert_scale_abs_f32.argtypes = [fp, up, ctypes.c_size_t, ctypes.c_float]
    lib.orc_convert_scale_abs_f64.argtypes = [dp, up, ctypes.c_size_t, ctypes.c_float]
    lib.orc_add_weighted_f32.argtypes = [fp, ctypes.c_float, fp, ctypes.c_float, fp, ctypes.c_size_t]
    lib.orc_add_weighted_f64.argtypes = [dp, ctypes.c_double, dp, ctypes.c_double, dp, ctypes.c_size_t]
    lib.orc_sepblur_f32_variant.argtypes = [fp, fp, fp, ci, ci, ci, fp, ci, fp, ci, ci, ci]
    lib.orc_remap_bilinear_f64_fma.argtypes = [dp, dp, ci, ci, ci, fp, fp]
    lib.orc_convert_scale_abs_f64_dbl.argtypes = [dp, up, ctypes.c_size_t, ctypes.c_float]
    _LIB = lib
    return lib


# Which restated OpenCV accumulation form the cv2 stand-ins below use.  The ORACLE is the default (all zeros); tests
# switch the other forms in through `opencv_variant` to measure the spread around it (crt_oracle.c, "ALTERNATIVE
# ACCUMULATION FORMS").  blur_row / blur_col: modes of orc_sepblur_f32_variant; remap_fma: contracted bilinear sum for
# CV_64F images; csa_double: convertScaleAbs of a CV_64F image multiplied in double (scalar tail of cvtabs_32f);
# resize_fma: the INTER_LINEAR passes contracted, the exact-2x mean summed pairwise (fast bloom, grain upsample).
VARIANT = {"blur_row": 0, "blur_col": 0, "remap_fma": 0, "csa_double": 0, "resize_fma": 0}
OPENCV_VARIANTS = {
    "oracle (RowFilter fma | ColumnFilter fma)": {},
    "SymmColumnFilter fma": {"blur_col": 1},
    "SymmColumnFilter mul+add": {"blur_col": 2, "blur_row": 1},
    "ColumnFilter mul+add (SSE baseline)": {"blur_col": 3, "blur_row": 1},
    "SymmRowSmall + SymmColumn fma": {"blur_row": 2, "blur_col": 1},
    "SymmRowSmall + SymmColumn mul+add": {"blur_row": 3, "blur_col": 2},
    "remap contracted (fma)": {"remap_fma": 1},
    "convertScaleAbs in double": {"csa_double": 1},
    "resize contracted (fma), pairwise 2x2 mean": {"resize_fma": 1},
    "all alternatives at once": {"blur_row": 3, "blur_col": 2, "remap_fma": 1, "csa_double": 1, "resize_fma": 1},
}


class opencv_variant:
    """with opencv_variant(blur_col=1): ... — run the restated chain with another of OpenCV's accumulation forms."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = dict(VARIANT)
        VARIANT.update(self.kw)
        return self

    def __exit__(self, *exc):
        VARIANT.clear()
        VARIANT.update(self.old)
        return False


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# --------------------------------------------------------------------------------------
# OpenCV restatements (parity unpinned)
# --------------------------------------------------------------------------------------

def gaussian_kernel(ksize: int, sigma: float) -> np.ndarray:
    """cv::getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0 (OpenCV 4.x bit-exact form):
    taps exp(-x^2/(2 sigma^2)) in double, the outer taps summed first, doubled, plus the centre
    1; normalised by one reciprocal; then narrowed to float.  Used by ref:234 and ref:610/780."""
    n = int(ksize)
    assert n >= 1 and n % 2 == 1 and sigma > 0.0
    n2 = (n - 1) // 2
    scale2x = np.float64(-0.125) / (np.float64(sigma) * np.float64(sigma))
    vals = np.empty(n2, dtype=np.float64)
    s = np.float64(0.0)
    x = 1 - n
    for i in range(n2):
        t = np.exp(np.float64(x * x) * scale2x)
        vals[i] = t
        s = s + t
        x += 2
    s = s * np.float64(2.0)
    s = s + np.float64(1.0)
    mul1 = np.float64(1.0) / s
    out = np.empty(n, dtype=np.float64)
    for i in range(n2):
        t = vals[i] * mul1
        out[i] = t
        out[n - 1 - i] = t
    out[n2] = mul1
    return out.astype(np.float32)


def gaussian_blur(src: np.ndarray, ksize: Tuple[int, int], sigma_x: float, sigma_y: float = 0.0) -> np.ndarray:
    """cv2.GaussianBlur(src, (kw,kh), sigmaX, sigmaY, borderType=BORDER_REPLICATE) for CV_32F.
    sigmaY<=0 takes sigmaX; a 1-tap axis gets the kernel [1]; ksize (1,1) is a copy."""
    kw, kh = int(ksize[0]), int(ksize[1])
    src = np.ascontiguousarray(src, dtype=np.float32)
    if kw == 1 and kh == 1:
        return src.copy()
    sx = float(sigma_x)
    sy = float(sigma_y) if sigma_y > 0 else sx
    kx = gaussian_kernel(kw, sx) if kw > 1 else np.ones(1, np.float32)
    if kh == kw and abs(sx - sy) < np.finfo(np.float64).eps:
        ky = kx
    else:
        ky = gaussian_kernel(kh, sy) if kh > 1 else np.ones(1, np.float32)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty_like(src)
    tmp = np.empty_like(src)
    if VARIANT["blur_row"] or VARIANT["blur_col"]:
        _lib().orc_sepblur_f32_variant(_fp(src), _fp(dst), _fp(tmp), h, w, cn, _fp(kx), len(kx), _fp(ky), len(ky),
                                       VARIANT["blur_row"], VARIANT["blur_col"])
    else:
        _lib().orc_sepblur_f32(_fp(src), _fp(dst), _fp(tmp), h, w, cn, _fp(kx), len(kx), _fp(ky), len(ky))
    return dst


def gaussian_blur_slow(src: np.ndarray, kx: np.ndarray, ky: np.ndarray) -> np.ndarray:
    """Pure-numpy twin of orc_sepblur_f32 (fma emulated through float64: the product of two
    floats is exact in double; only a double-rounding tie could differ).  Cross-check only."""
    src = np.asarray(src, dtype=np.float32)
    h, w = src.shape[:2]
    rx, ry = len(kx) // 2, len(ky) // 2
    xs = np.arange(w)
    acc = np.zeros(src.shape, np.float32)
    for k in range(len(kx)):
        xx = np.clip(xs + k - rx, 0, w - 1)
        acc = (src[:, xx].astype(np.float64) * np.float64(kx[k]) + acc.astype(np.float64)).astype(np.float32)
    ys = np.arange(h)
    out = np.zeros(src.shape, np.float32)
    for k in range(len(ky)):
        yy = np.clip(ys + k - ry, 0, h - 1)
        out = (acc[yy].astype(np.float64) * np.float64(ky[k]) + out.astype(np.float64)).astype(np.float32)
    return out


def remap_quantise(map_x: np.ndarray, map_y: np.ndarray):
    """Integer tap origin and packed 5+5-bit fraction of cv2.remap's fixed-point map."""
    mx = np.ascontiguousarray(map_x, np.float32)
    my = np.ascontiguousarray(map_y, np.float32)
    n = mx.size
    ix = np.empty(mx.shape, np.int32)
    iy = np.empty(mx.shape, np.int32)
    fxy = np.empty(mx.shape, np.int32)
    ip = ctypes.POINTER(ctypes.c_int32)
    _lib().orc_remap_quantise(_fp(mx), _fp(my), n, ix.ctypes.data_as(ip), iy.ctypes.data_as(ip), fxy.ctypes.data_as(ip))
    return ix, iy, fxy


def remap_bilinear(img: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(img, map_x, map_y, INTER_LINEAR, BORDER_CONSTANT, 0) for CV_32F / CV_64F."""
    mx = np.ascontiguousarray(map_x, np.float32)
    my = np.ascontiguousarray(map_y, np.float32)
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    if img.dtype == np.float64:
        src = np.ascontiguousarray(img)
        dst = np.empty_like(src)
        (_lib().orc_remap_bilinear_f64_fma if VARIANT["remap_fma"] else _lib().orc_remap_bilinear_f64)(_dp(src), _dp(dst), h, w, cn, _fp(mx), _fp(my))
    else:
        src = np.ascontiguousarray(img, np.float32)
        dst = np.empty_like(src)
        _lib().orc_remap_bilinear_f32(_fp(src), _fp(dst), h, w, cn, _fp(mx), _fp(my))
    return dst


def resize(img: np.ndarray, dsize: Tuple[int, int], interpolation: str) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=INTER_NEAREST|INTER_LINEAR) for CV_32F; INTER_LINEAR also for
    CV_64F (the float64 persistence state of a promoted chain, ref:690)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    if interpolation == "linear" and img.dtype == np.float64:
        src = np.ascontiguousarray(img)
        sh, sw = src.shape[:2]
        dst = np.empty((dh, dw) + src.shape[2:], np.float64)
        rc = _lib().orc_resize_linear_f64(_dp(src), sh, sw, _dp(dst), dh, dw, 1 if src.ndim == 2 else src.shape[2])
        assert rc == 0
        return dst
    src = np.ascontiguousarray(img, np.float32)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((dh, dw) + src.shape[2:], np.float32)
    if interpolation == "nearest":
        _lib().orc_resize_nearest_f32(_fp(src), sh, sw, _fp(dst), dh, dw, cn)
    elif interpolation == "linear":
        if VARIANT["resize_fma"]:
            rc = _lib().orc_resize_linear_f32_variant(_fp(src), sh, sw, _fp(dst), dh, dw, cn, 1)
        else:
            rc = _lib().orc_resize_linear_f32(_fp(src), sh, sw, _fp(dst), dh, dw, cn)
        assert rc == 0
    else:
        raise ValueError(interpolation)
    return dst


def convert_scale_abs(img: np.ndarray, alpha: float = 255.0) -> np.ndarray:
    """cv2.convertScaleAbs(img, alpha=255.0, beta=0) → uint8 (ref:696, :1098, :1124)."""
    up = ctypes.POINTER(ctypes.c_uint8)
    out = np.empty(img.shape, np.uint8)
    if img.dtype == np.float64:
        a = np.ascontiguousarray(img)
        (_lib().orc_convert_scale_abs_f64_dbl if VARIANT["csa_double"] else _lib().orc_convert_scale_abs_f64)(_dp(a), out.ctypes.data_as(up), a.size, alpha)
    else:
        a = np.ascontiguousarray(img, np.float32)
        _lib().orc_convert_scale_abs_f32(_fp(a), out.ctypes.data_as(up), a.size, alpha)
    return out


def add_weighted(a: np.ndarray, alpha: float, b: np.ndarray, beta: float) -> np.ndarray:
    """cv2.addWeighted(a, alpha, b, beta, 0.0) (ref:693).  Both inputs must share a dtype in
    OpenCV; a mixed f32/f64 pair (state from a promoted chain vs an unpromoted frame) is
    widened to float64 here."""
    if a.dtype == np.float64 or b.dtype == np.float64:
        a64 = np.ascontiguousarray(a, np.float64)
        b64 = np.ascontiguousarray(b, np.float64)
        out = np.empty_like(a64)
        _lib().orc_add_weighted_f64(_dp(a64), alpha, _dp(b64), beta, _dp(out), a64.size)
        return out
    a32 = np.ascontiguousarray(a, np.float32)
    b32 = np.ascontiguousarray(b, np.float32)
    out = np.empty_like(a32)
    _lib().orc_add_weighted_f32(_fp(a32), alpha, _fp(b32), beta, _fp(out), a32.size)
    return out


# --------------------------------------------------------------------------------------
# Effect primitives (ref:207-348) — numpy stages follow the reference expression by
# expression so that dtype promotion and rounding order are identical.
# --------------------------------------------------------------------------------------

def shift_channel(arr: np.ndarray, dx: int, dy: int) -> np.ndarray:
    """ref:207-210 — wrap-around roll: out[y, x] = arr[(y-dy) % H, (x-dx) % W]."""
    if dx == 0 and dy == 0:
        return arr
    return np.roll(np.roll(arr, dy, axis=0), dx, axis=1)


def make_scanline_mask_dynamic(h: int, strength: float, period_px: float, phase_px: float) -> np.ndarray:
    """ref:213-217 — float32 length-h row gain 1 - s*0.5*(1+sin(2pi/P*(y+phase)))."""
    y = np.arange(h, dtype=np.float32)
    s = 0.5 * (1.0 + np.sin((2.0 * np.pi / max(1e-6, period_px)) * (y + phase_px)))
    return 1.0 - strength * s


def triad_ksize(softness_px: float) -> int:
    """ref:231-233 — python round() is banker's rounding."""
    s = float(max(0.0, softness_px))
    return max(3, int(round(s * 3)) * 2 + 1)


def make_triad_row(w: int, strength: float) -> np.ndarray:
    """ref:221-229 — one row (1, w, 3) of the unsoftened aperture mask."""
    x = np.arange(w)[None, :]
    m0 = (x % 3 == 0).astype(np.float32)
    m1 = (x % 3 == 1).astype(np.float32)
    m2 = (x % 3 == 2).astype(np.float32)
    base = 1.0 - float(strength)
    r = base + float(strength) * m0
    g = base + float(strength) * m1
    b = base + float(strength) * m2
    return np.stack([r, g, b], axis=2).astype(np.float32)


def make_triad_mask(h: int, w: int, strength: float, softness_px: float = 0.0) -> np.ndarray:
    """ref:220-235 — H x W x 3 float32; rows are identical (repeat at ref:230, horizontal-only
    blur at ref:234)."""
    mask = make_triad_row(w, strength)
    mask = np.repeat(mask, h, axis=0)
    s = float(max(0.0, softness_px))
    if s > 0.0:
        k = triad_ksize(s)
        mask = gaussian_blur(mask, (k, 1), s, 0.0)
    return mask.astype(np.float32)


def triad_luts(gamma: float):
    """ref:246-249, :260 — the two 1025-entry float32 LUTs."""
    g = float(gamma)
    lut_x = np.linspace(0.0, 1.0, 1025, dtype=np.float32)
    lut_g = np.power(lut_x, g, dtype=np.float32)
    lut_inv = np.power(lut_x, 1.0 / g, dtype=np.float32)
    return lut_g, lut_inv


def apply_triad_mask(img: np.ndarray, mask: np.ndarray, gamma: float = 2.2, preserve_luma: bool = True) -> np.ndarray:
    """ref:238-263."""
    g = float(gamma)
    if (not preserve_luma) and abs(g - 1.0) < 1e-3:
        return np.clip(img * mask, 0.0, 1.0)
    if g <= 0.0:
        return np.clip(img * mask, 0.0, 1.0)
    lut_size = 1024
    scale = float(lut_size)
    lut_g, lut_inv = triad_luts(g)
    idx = np.clip((np.clip(img, 0.0, 1.0) * scale).astype(np.int32), 0, lut_size)
    lin = lut_g[idx]
    out_lin = lin * mask
    if preserve_luma:
        w_r, w_g, w_b = 0.2126, 0.7152, 0.0722
        y_before = w_r * lin[:, :, 0] + w_g * lin[:, :, 1] + w_b * lin[:, :, 2]
        y_after = w_r * out_lin[:, :, 0] + w_g * out_lin[:, :, 1] + w_b * out_lin[:, :, 2]
        ratio = y_before / np.maximum(y_after, 1e-6)
        ratio = np.clip(ratio, 0.5, 2.0)
        out_lin = out_lin * ratio[:, :, None]
    idx2 = np.clip((np.clip(out_lin, 0.0, 1.0) * scale).astype(np.int32), 0, lut_size)
    out = lut_inv[idx2]
    return np.clip(out, 0.0, 1.0)


def make_vignette(h: int, w: int, strength: float) -> np.ndarray:
    """ref:266-276 — float64 H x W."""
    yy, xx = np.mgrid[0:h, 0:w]
    cx = (w - 1) / 2.0
    cy = (h - 1) / 2.0
    rx = max(1.0, w / 2.0)
    ry = max(1.0, h / 2.0)
    nx = (xx - cx) / rx
    ny = (yy - cy) / ry
    r2 = nx * nx + ny * ny
    return 1.0 - strength * np.clip(r2, 0.0, 1.0)


def apply_color_adjustments(img, brightness, contrast, gamma, saturation, temperature):
    """ref:279-305 — gates are exact != comparisons; temperature writes in place."""
    if saturation != 1.0:
        luma = 0.2126 * img[:, :, 0] + 0.7152 * img[:, :, 1] + 0.0722 * img[:, :, 2]
        img = np.clip(luma[:, :, None] + (img - luma[:, :, None]) * float(saturation), 0.0, 1.0)
    if temperature != 0.0:
        t = float(temperature)
        r_gain = float(np.clip(1.0 + 0.5 * t, 0.5, 1.5))
        b_gain = float(np.clip(1.0 - 0.5 * t, 0.5, 1.5))
        img[:, :, 0] = np.clip(img[:, :, 0] * r_gain, 0.0, 1.0)
        img[:, :, 2] = np.clip(img[:, :, 2] * b_gain, 0.0, 1.0)
    if brightness != 0.0 or contrast != 1.0:
        img = np.clip((img - 0.5) * float(contrast) + 0.5 + float(brightness), 0.0, 1.0)
    if gamma != 1.0 and gamma > 0.0:
        inv_g = 1.0 / float(gamma)
        img = np.clip(np.power(img, inv_g, dtype=np.float32), 0.0, 1.0)
    return img


def make_scanline_mask_2d(h, w, strength, period_px, phase_px, angle_deg, thickness) -> np.ndarray:
    """ref:308-328 — computed in float64, returned float32."""
    if strength <= 0.0:
        return np.ones((h, w), dtype=np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    theta = np.deg2rad(float(angle_deg))
    slanted = yy + np.tan(theta) * xx
    omega = 2.0 * np.pi / max(1e-6, float(period_px))
    s = 0.5 * (1.0 + np.sin(omega * (slanted + float(phase_px))))
    sharp = np.clip(float(thickness), 0.1, 4.0)
    s_shaped = np.power(s, 1.0 / sharp)
    mask = 1.0 - float(strength) * s_shaped
    return mask.astype(np.float32)


def barrel_maps(h: int, w: int, strength: float):
    """ref:335-346 — the float32 sampling maps handed to cv2.remap."""
    s = float(strength)
    cx = (w - 1) / 2.0
    cy = (h - 1) / 2.0
    x = (np.arange(w, dtype=np.float32) - cx) / max(1.0, cx)
    y = (np.arange(h, dtype=np.float32) - cy) / max(1.0, cy)
    xv, yv = np.meshgrid(x, y)
    r2 = xv * xv + yv * yv
    k = s * 0.5
    factor = 1.0 + k * r2
    map_x = (xv * factor * cx + cx).astype(np.float32)
    map_y = (yv * factor * cy + cy).astype(np.float32)
    return map_x, map_y


def apply_barrel_warp(img: np.ndarray, strength: float) -> np.ndarray:
    """ref:331-348."""
    s = float(strength)
    if s == 0.0:
        return img
    h, w = img.shape[:2]
    map_x, map_y = barrel_maps(h, w, s)
    return remap_bilinear(img, map_x, map_y)


def bloom_ksize(bloom_sigma: float) -> int:
    """ref:609 / :779 — python round() (banker's): sigma 3 -> 19, 1.2 -> 9, 1.5 -> 9."""
    return max(1, int(round(bloom_sigma * 3)) * 2 + 1)


def _overlay(img, ov, h, w):
    """ref:588-598 / :653-663, including the Pillow bilinear resize of an overlay of another size (ref:593-594)."""
    if ov.dtype != np.uint8:
        ov = np.clip(ov, 0, 255).astype(np.uint8)
    if ov.shape[0] != h or ov.shape[1] != w:
        from PIL import Image
        ov = np.asarray(Image.fromarray(ov, mode="RGBA").resize((w, h), Image.BILINEAR))
    alpha = (ov[:, :, 3:4].astype(np.float32)) / 255.0
    rgb = ov[:, :, :3].astype(np.float32) / 255.0
    return np.clip(img * (1.0 - alpha) + rgb * alpha, 0.0, 1.0)


def glitch_offsets_render(h2: int, w2: int, scanline_phase_px: float, glitch_amp_px: int, glitch_height_frac: float):
    """ref:838-855 — (y0, per-pixel int32 x-offsets of the bottom band) of the render variant."""
    y0 = max(0, min(h2, h2 - int(h2 * glitch_height_frac)))
    if y0 >= h2:
        return y0, None
    num_rows = h2 - y0
    seed = (int(abs(float(scanline_phase_px)) * 2.0) + (w2 << 10) + (h2 << 1)) & 0xFFFFFFFF
    rng = np.random.default_rng(seed)
    seg_len = max(8, min(32, w2 // 120 if w2 >= 120 else 8))
    num_segs = (w2 + seg_len - 1) // seg_len
    rows_idx = np.arange(num_rows, dtype=np.float32)
    amp_rows = float(glitch_amp_px) * (1.0 - (rows_idx / max(1.0, float(num_rows))))
    seg_offsets = rng.standard_normal((num_rows, num_segs)).astype(np.float32) * (amp_rows[:, None] * 0.7)
    base_rw = rng.standard_normal(num_rows).astype(np.float32)
    base = np.cumsum(base_rw) * 0.1
    base = np.clip(base, -amp_rows * 0.4, amp_rows * 0.4)
    seg_index = (np.arange(w2, dtype=np.int32) // int(seg_len)).astype(np.int32)
    offs_pp = base[:, None] + seg_offsets[np.arange(num_rows)[:, None], seg_index[None, :]]
    return y0, np.rint(offs_pp).astype(np.int32)


def glitch_offsets_preview(h2: int, w2: int, scanline_phase_px: float, glitch_amp_px: int, glitch_height_frac: float):
    """ref:667-682 — (y0, per-row int32 x-offsets) of the preview variant."""
    y0 = max(0, min(h2, h2 - int(h2 * glitch_height_frac)))
    if y0 >= h2:
        return y0, None
    num_rows = h2 - y0
    seed = (int(abs(float(scanline_phase_px)) * 0.05) + (w2 << 10) + (h2 << 1)) & 0xFFFFFFFF
    rng = np.random.default_rng(seed)
    rows_idx = np.arange(num_rows, dtype=np.float32)
    amp_rows = np.asarray(float(glitch_amp_px) * np.exp(-3.0 * (rows_idx / max(1.0, float(num_rows)))), dtype=np.float32)
    base = rng.normal(loc=0.0, scale=0.5, size=num_rows).astype(np.float32)
    base = np.clip(base, -1.0, 1.0)
    jump_mask = rng.random(num_rows).astype(np.float32) < 0.03
    jump_sign = rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=num_rows)
    base = base + jump_mask * jump_sign
    offs_row = np.clip(base * amp_rows, -amp_rows, amp_rows)
    return y0, np.rint(offs_row).astype(np.int32)[:, None]


def _apply_glitch(img, y0, offs):
    """ref:680-685 / :852-858 — horizontal wrap-gather of the bottom band, in place."""
    if offs is None:
        return img
    h2, w2 = img.shape[0], img.shape[1]
    bottom = img[y0:, :, :]
    x = np.arange(w2, dtype=np.int32)[None, :]
    xi = (x + offs) % w2
    idx = np.broadcast_to(xi[:, :, None], bottom.shape)
    img[y0:, :, :] = np.take_along_axis(bottom, idx, axis=1)
    return img


def _noise(h, w, grain_size, noise_plane):
    """ref:637-645 / :807-815 with the N(0,1) draw injected (cv2.randn is unreproducible)."""
    if grain_size and grain_size > 1:
        gh = max(1, h // int(grain_size))
        gw = max(1, w // int(grain_size))
        small = np.ascontiguousarray(noise_plane, np.float32)
        assert small.shape == (gh, gw), f"noise_plane must be {(gh, gw)} for grain_size {grain_size}"
        return resize(small, (w, h), "linear")
    noise = np.ascontiguousarray(noise_plane, np.float32)
    assert noise.shape == (h, w)
    return noise


def _static_chain(frame, scanline_strength, triad_mask, triad_gamma, triad_preserve_luma, aberration_px,
                  bloom_sigma, bloom_strength, bloom_threshold, noise_strength, vignette_mask,
                  scanline_period_px, scanline_phase_px, fast_bloom, pixel_size, time_sec, brightness,
                  contrast, gamma, saturation, temperature, flicker_strength, flicker_hz, grain_size,
                  scanline_angle, scanline_thickness, warp_strength, text_overlay_rgba, text_overlay_after,
                  noise_plane, stop_before_warp=False):
    """Shared body of ref:566-663 and ref:735-834 (identical statement for statement)."""
    h, w = frame.shape[0], frame.shape[1]
    img = frame.astype(np.float32) / 255.0
    if aberration_px != 0:
        r = shift_channel(img[:, :, 0], aberration_px, 0)
        g = img[:, :, 1]
        b = shift_channel(img[:, :, 2], -aberration_px, 0)
        img = np.stack([r, g, b], axis=2)
    if pixel_size > 1:
        sw = max(1, w // int(pixel_size))
        sh = max(1, h // int(pixel_size))
        img = resize(img, (sw, sh), "nearest")
        img = resize(img, (w, h), "nearest")
    img = apply_color_adjustments(img, brightness, contrast, gamma, saturation, temperature)
    if text_overlay_rgba is not None and not text_overlay_after:
        img = _overlay(img, text_overlay_rgba, h, w)
    if bloom_strength > 0.0 and (bloom_sigma > 0.0 or fast_bloom):
        src = img
        if bloom_threshold > 0.0:
            thr = float(min(0.99, max(0.0, bloom_threshold)))
            src = np.clip((img - thr) / max(1e-6, (1.0 - thr)), 0.0, 1.0)
        if fast_bloom:
            ds = resize(src, (max(1, w // 2), max(1, h // 2)), "linear")
            blurf = resize(ds, (w, h), "linear")
        else:
            k = bloom_ksize(bloom_sigma)
            blurf = gaussian_blur(src, (k, k), bloom_sigma, bloom_sigma)
        img = np.clip(img + bloom_strength * blurf, 0.0, 1.0)
    if triad_mask is not None:
        img = apply_triad_mask(img, triad_mask, triad_gamma, triad_preserve_luma)
    if scanline_strength > 0.0:
        if scanline_angle == 0.0 and scanline_thickness == 1.0:
            sl = make_scanline_mask_dynamic(h, scanline_strength, scanline_period_px, scanline_phase_px)
            img = np.clip(img * sl[:, None, None], 0.0, 1.0)
        else:
            sl2d = make_scanline_mask_2d(h, w, scanline_strength, scanline_period_px, scanline_phase_px,
                                         scanline_angle, scanline_thickness)
            img = np.clip(img * sl2d[:, :, None], 0.0, 1.0)
    if vignette_mask is not None:
        img = np.clip(img * vignette_mask[:, :, None], 0.0, 1.0)
    if flicker_strength > 0.0 and flicker_hz > 0.0:
        factor = 1.0 + 0.25 * float(flicker_strength) * np.sin(2.0 * np.pi * float(flicker_hz) * float(time_sec))
        img = np.clip(img * factor, 0.0, 1.0)
    if noise_strength > 0.0:
        noise = _noise(h, w, grain_size, noise_plane)
        noise = noise * (noise_strength / 255.0)
        img = np.clip(img + noise[:, :, None], 0.0, 1.0)
    if stop_before_warp:
        return img
    if warp_strength != 0.0:
        img = apply_barrel_warp(img, warp_strength)
    if text_overlay_rgba is not None and text_overlay_after:
        img = _overlay(img, text_overlay_rgba, h, w)
    return img


def apply_static_effects(frame, scanline_strength, triad_mask, triad_gamma, triad_preserve_luma, aberration_px,
                         bloom_sigma, bloom_strength, bloom_threshold, noise_strength, vignette_mask,
                         scanline_period_px, scanline_phase_px, fast_bloom, pixel_size, glitch_amp_px,
                         glitch_height_frac, time_sec=0.0, brightness=0.0, contrast=1.0, gamma=1.0,
                         saturation=1.0, temperature=0.0, flicker_strength=0.0, flicker_hz=0.0, grain_size=1,
                         scanline_angle=0.0, scanline_thickness=1.0, warp_strength=0.0, text_overlay_rgba=None,
                         text_overlay_after=True, *, noise_plane=None, stop_before_warp=False) -> np.ndarray:
    """ref:702-861 — stateless render-path chain; returns the float image (float64 once the
    vignette or flicker stage has run).  `noise_plane` / `stop_before_warp` are oracle-only."""
    img = _static_chain(frame, scanline_strength, triad_mask, triad_gamma, triad_preserve_luma, aberration_px,
                        bloom_sigma, bloom_strength, bloom_threshold, noise_strength, vignette_mask,
                        scanline_period_px, scanline_phase_px, fast_bloom, pixel_size, time_sec, brightness,
                        contrast, gamma, saturation, temperature, flicker_strength, flicker_hz, grain_size,
                        scanline_angle, scanline_thickness, warp_strength, text_overlay_rgba,
                        text_overlay_after, noise_plane, stop_before_warp)
    if stop_before_warp:
        return img
    if glitch_amp_px > 0 and glitch_height_frac > 0.0:
        y0, offs = glitch_offsets_render(img.shape[0], img.shape[1], scanline_phase_px, glitch_amp_px, glitch_height_frac)
        img = _apply_glitch(img, y0, offs)
    return img


def apply_crt_effect(frame, scanline_strength, triad_mask, triad_gamma, triad_preserve_luma, aberration_px,
                     bloom_sigma, bloom_strength, bloom_threshold, noise_strength, vignette_mask, persistence,
                     state_prev, scanline_period_px, scanline_phase_px, fast_bloom, pixel_size, glitch_amp_px=0,
                     glitch_height_frac=0.0, time_sec=0.0, brightness=0.0, contrast=1.0, gamma=1.0,
                     saturation=1.0, temperature=0.0, flicker_strength=0.0, flicker_hz=0.0, grain_size=1,
                     scanline_angle=0.0, scanline_thickness=1.0, warp_strength=0.0, text_overlay_rgba=None,
                     text_overlay_after=True, *, noise_plane=None):
    """ref:531-699 — stateful preview-path chain; returns (uint8 frame, float state)."""
    h, w = frame.shape[0], frame.shape[1]
    img = _static_chain(frame, scanline_strength, triad_mask, triad_gamma, triad_preserve_luma, aberration_px,
                        bloom_sigma, bloom_strength, bloom_threshold, noise_strength, vignette_mask,
                        scanline_period_px, scanline_phase_px, fast_bloom, pixel_size, time_sec, brightness,
                        contrast, gamma, saturation, temperature, flicker_strength, flicker_hz, grain_size,
                        scanline_angle, scanline_thickness, warp_strength, text_overlay_rgba,
                        text_overlay_after, noise_plane)
    if glitch_amp_px > 0 and glitch_height_frac > 0.0:
        y0, offs = glitch_offsets_preview(img.shape[0], img.shape[1], scanline_phase_px, glitch_amp_px, glitch_height_frac)
        img = _apply_glitch(img, y0, offs)
    if state_prev is not None and persistence > 0.0:
        if state_prev.shape != img.shape:
            prev_arr = resize(state_prev, (w, h), "linear")
        else:
            prev_arr = state_prev
        img = add_weighted(prev_arr, float(persistence), img, float(1.0 - persistence))
    out = convert_scale_abs(img, 255.0)
    return out, img


def persistence_blend(prev_state: Optional[np.ndarray], static_img: np.ndarray, persistence: float):
    """ref:1086-1098 — render-path in-order commit: returns (blended state, uint8 frame).
    The first frame (prev_state None) passes through unblended."""
    if prev_state is not None and persistence > 0.0:
        blended = np.clip(persistence * prev_state + (1.0 - persistence) * static_img, 0.0, 1.0)
    else:
        blended = static_img
    return blended, convert_scale_abs(blended, 255.0)


def process_frames(frames, params: dict, fps: float, scanline_speed_px_s: float, persistence: float,
                   triad_strength: float, triad_softness: float, vignette_strength: float,
                   noise_planes=None, first_index: int = 0, prev_state=None):
    """The hot slice of process_video (ref:919-920, :1037-1131) over in-memory frames:
    masks built once, phase = i/fps*speed (ref:1043), time_sec = i/fps (ref:1064), in-order
    persistence IIR and uint8 quantise.  `params` holds the remaining apply_static_effects
    keywords.  Returns (list of uint8 frames, final float state)."""
    h, w = frames[0].shape[:2]
    triad_mask = make_triad_mask(h, w, triad_strength, triad_softness) if triad_strength > 0.0 else None
    vignette_mask = make_vignette(h, w, vignette_strength) if vignette_strength > 0.0 else None
    outs = []
    for j, frame in enumerate(frames):
        i = first_index + j
        phase = (i / float(fps)) * scanline_speed_px_s
        static = apply_static_effects(
            frame, params["scanline_strength"], triad_mask, float(params["triad_gamma"]),
            bool(params["triad_preserve_luma"]), params["aberration_px"], params["bloom_sigma"],
            params["bloom_strength"], float(params.get("bloom_threshold", 0.0)), params["noise_strength"],
            vignette_mask, params["scanline_period_px"], phase, params["fast_bloom"], params["pixel_size"],
            int(params.get("glitch_amp_px", 0)), float(params.get("glitch_height_frac", 0.0)),
            time_sec=(i / float(fps)), brightness=float(params.get("brightness", 0.0)),
            contrast=float(params.get("contrast", 1.0)), gamma=float(params.get("gamma", 1.0)),
            saturation=float(params.get("saturation", 1.0)), temperature=float(params.get("temperature", 0.0)),
            flicker_strength=float(params.get("flicker_strength", 0.0)),
            flicker_hz=float(params.get("flicker_hz", 0.0)), grain_size=int(params.get("grain_size", 1)),
            scanline_angle=float(params.get("scanline_angle", 0.0)),
            scanline_thickness=float(params.get("scanline_thickness", 1.0)),
            warp_strength=float(params.get("warp_strength", 0.0)),
            text_overlay_rgba=params.get("text_overlay_rgba"), text_overlay_after=bool(params.get("text_overlay_after", True)),
            noise_plane=None if noise_planes is None else noise_planes[j])
        prev_state, out = persistence_blend(prev_state, static, persistence)
        outs.append(out)
    return outs, prev_state
