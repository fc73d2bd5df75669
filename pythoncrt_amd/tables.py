"""Host-side parameter tables for libcrtfx (product code — never imports oracle/).

Everything here is O(W + H) or O(1) per frame: Gaussian taps, the single distinct row of the
triad mask, the two 1025-entry LUTs, vignette / warp axis vectors, scanline row gains, flicker
factor, pixelate index maps.  The per-pixel work lives in the HIP kernels.  Where the reference
(crt_filter.py, `ref:LINE`) computes a table with numpy, the same numpy expression is used so
the values are this machine's numpy values (np.sin / np.power are not bit-portable across CPUs).

Restated reference expressions.  Three helpers necessarily repeat lines of crt_filter.py (PythonCRT, GPL-3.0 — see NOTICE at the
repository root), because the product must draw the same random numbers in the same order and evaluate the same numpy expression
tree to be bit-exact with it: `glitch_offsets_preview` (ref:670-679) and `glitch_offsets_render_segments` (ref:841-850) — the PCG64
seed formula and the order of the Generator calls; `grade_lut` (ref:292-304) — the float32 / float64 expression order.  Each cites
its lines; nothing else in this package is taken from the reference's text.  (The 2-D scanline mask, ref:317-328, is written on the
device by crtfx_scanline_plane; its numpy form lives in the oracle only.)
"""
from __future__ import annotations

import ctypes

import numpy as np


def bloom_ksize(bloom_sigma: float) -> int:
    """ref:609 — k = max(1, round(3 sigma) * 2 + 1), python (banker's) round."""
    return max(1, int(round(bloom_sigma * 3)) * 2 + 1)


def triad_ksize(softness_px: float) -> int:
    """ref:231-233."""
    s = float(max(0.0, softness_px))
    return max(3, int(round(s * 3)) * 2 + 1)


def gaussian_taps(ksize: int, sigma: float) -> np.ndarray:
    """cv::getGaussianKernel(ksize, sigma, CV_32F), sigma > 0: exp(-x^2 / 2 sigma^2) in double
    with x counted in half-pixels, outer taps summed first then doubled plus the centre one,
    one reciprocal to normalise, narrowed to float32."""
    n = int(ksize)
    if n < 1 or n % 2 == 0 or not sigma > 0.0:
        raise ValueError(f"gaussian_taps({ksize}, {sigma})")
    half = (n - 1) // 2
    scale = np.float64(-0.125) / (np.float64(sigma) * np.float64(sigma))
    xs = np.arange(1 - n, 0, 2, dtype=np.int64)                  # 1-n, 3-n, ..., -2
    outer = np.exp((xs * xs).astype(np.float64) * scale)
    total = np.float64(0.0)
    for t in outer:                                              # sequential sum, as OpenCV does
        total = total + t
    total = total * np.float64(2.0) + np.float64(1.0)
    inv = np.float64(1.0) / total
    taps = np.empty(n, np.float64)
    taps[:half] = outer * inv
    taps[half] = inv
    taps[half + 1:] = taps[:half][::-1]
    return taps.astype(np.float32)


def triad_row(lib, w: int, strength: float, softness_px: float = 0.0) -> np.ndarray:
    """The one distinct row (w, 3) float32 of make_triad_mask (ref:220-235): the mask is a
    row repeated h times (ref:230) and the softening blur is horizontal-only (ref:234)."""
    x = np.arange(w)[None, :]
    base = 1.0 - float(strength)
    chans = [(base + float(strength) * (x % 3 == c).astype(np.float32)) for c in range(3)]
    row = np.ascontiguousarray(np.stack(chans, axis=2).astype(np.float32)[0])      # (w, 3)
    s = float(max(0.0, softness_px))
    if s > 0.0:
        taps = gaussian_taps(triad_ksize(s), s)
        out = np.empty_like(row)
        rc = lib.crtfx_host_blur_row(row.ctypes.data, out.ctypes.data, w, 3, taps.ctypes.data, len(taps))
        if rc != 0:
            raise RuntimeError(f"crtfx_host_blur_row failed: {rc}")
        row = out
    return row


def triad_luts(gamma: float):
    """ref:246-249, :260."""
    g = float(gamma)
    lut_x = np.linspace(0.0, 1.0, 1025, dtype=np.float32)
    return (np.ascontiguousarray(np.power(lut_x, g, dtype=np.float32)),
            np.ascontiguousarray(np.power(lut_x, 1.0 / g, dtype=np.float32)))


def triad_uses_lut(gamma: float, preserve_luma: bool) -> bool:
    """ref:240-245 — the plain-multiply shortcuts."""
    g = float(gamma)
    if (not preserve_luma) and abs(g - 1.0) < 1e-3:
        return False
    return g > 0.0


def vignette_axes(h: int, w: int):
    """ref:267-274 separated per axis: nx^2 (W,) and ny^2 (H,), float64.  The kernel forms
    r2 = nx2[x] + ny2[y] and v = 1 - strength * clip(r2, 0, 1) exactly as ref:274-275."""
    cx = (w - 1) / 2.0
    cy = (h - 1) / 2.0
    rx = max(1.0, w / 2.0)
    ry = max(1.0, h / 2.0)
    nx = (np.arange(w) - cx) / rx
    ny = (np.arange(h) - cy) / ry
    return np.ascontiguousarray(nx * nx), np.ascontiguousarray(ny * ny)


def vignette_full(h: int, w: int, strength: float) -> np.ndarray:
    """The H x W float64 array make_vignette returns (ref:266-276); only materialised when a
    caller asks the descriptor for its array form."""
    nx2, ny2 = vignette_axes(h, w)
    return 1.0 - strength * np.clip(nx2[None, :] + ny2[:, None], 0.0, 1.0)


def warp_axes(h: int, w: int):
    """ref:336-339 — float32 normalised axes and the centre."""
    cx = (w - 1) / 2.0
    cy = (h - 1) / 2.0
    x = (np.arange(w, dtype=np.float32) - cx) / max(1.0, cx)
    y = (np.arange(h, dtype=np.float32) - cy) / max(1.0, cy)
    return np.ascontiguousarray(x), np.ascontiguousarray(y), cx, cy


def scanline_rows(h: int, strength: float, period_px: float, phases) -> np.ndarray:
    """make_scanline_mask_dynamic (ref:213-217) for a batch of phases -> (B, h) float32.
    Row b equals the reference's mask for phase_px = phases[b] bit for bit: the phase is added
    as a float32 (a python float is a weak scalar next to the float32 `y`)."""
    y = np.arange(h, dtype=np.float32)[None, :]
    ph = np.asarray(phases, dtype=np.float64).astype(np.float32)[:, None]
    s = 0.5 * (1.0 + np.sin((2.0 * np.pi / max(1e-6, period_px)) * (y + ph)))
    return np.ascontiguousarray(1.0 - strength * s)


def scanline_rows_at(k: np.ndarray, strength: float, period_px: float) -> np.ndarray:
    """The expression of scanline_rows on given float32 values of (y + phase): used when those sums are exact integers
    (pipeline._scan_rows), so one table serves every frame of a batch."""
    k = np.asarray(k, dtype=np.float32)
    s = 0.5 * (1.0 + np.sin((2.0 * np.pi / max(1e-6, period_px)) * k))
    return np.ascontiguousarray(1.0 - strength * s)


def grade_lut(brightness: float, contrast: float, gamma: float, temperature: float) -> np.ndarray:
    """a1 + a4 for every uint8 code and channel when saturation == 1 (the only stage of apply_color_adjustments that
    mixes channels, ref:288-290): (3, 256) float32, lut[c, u] = value of channel c of a pixel whose sample is u after
    ref:569 (u / 255.0) and ref:292-304, computed with the reference's own expressions on a 1 x 256 x 3 image."""
    v = np.arange(256, dtype=np.uint8).astype(np.float32) / 255.0                         # ref:569
    img = np.ascontiguousarray(np.repeat(v[None, :, None], 3, axis=2))                    # 1 x 256 x 3
    if temperature != 0.0:                                                                # ref:292-297
        t = float(temperature)
        r_gain = float(np.clip(1.0 + 0.5 * t, 0.5, 1.5))
        b_gain = float(np.clip(1.0 - 0.5 * t, 0.5, 1.5))
        img[:, :, 0] = np.clip(img[:, :, 0] * r_gain, 0.0, 1.0)
        img[:, :, 2] = np.clip(img[:, :, 2] * b_gain, 0.0, 1.0)
    if brightness != 0.0 or contrast != 1.0:                                              # ref:299-300
        img = np.clip((img - 0.5) * float(contrast) + 0.5 + float(brightness), 0.0, 1.0)
    if gamma != 1.0 and gamma > 0.0:                                                      # ref:302-304
        img = np.clip(np.power(img, 1.0 / float(gamma), dtype=np.float32), 0.0, 1.0)
    assert img.dtype == np.float32
    return np.ascontiguousarray(img[0].T)                                                 # (3, 256)


def scanline_plane_scalars(period_px: float, angle_deg: float, thickness: float):
    """(omega, tan(theta), 1/sharp) of make_scanline_mask_2d exactly as ref:319-324 computes them: the scalar
    arguments of crtfx_scanline_plane."""
    theta = np.deg2rad(float(angle_deg))
    omega = 2.0 * np.pi / max(1e-6, float(period_px))
    sharp = np.clip(float(thickness), 0.1, 4.0)
    return float(omega), float(np.tan(theta)), float(1.0 / sharp)


def flicker_factor(flicker_strength: float, flicker_hz: float, time_sec: float) -> float:
    """ref:632."""
    return float(1.0 + 0.25 * float(flicker_strength) * np.sin(2.0 * np.pi * float(flicker_hz) * float(time_sec)))


def pixelate_maps(h: int, w: int, pixel_size: int):
    """Composite source index of the INTER_NEAREST down/up pair (ref:580-583):
    OpenCV's resizeNN maps dst index j to min(floor(j * (1 / (dst/src))), src-1), ratio in double."""
    def nn(n_dst, n_src):
        ifx = 1.0 / (n_dst / n_src)
        return np.minimum(np.floor(np.arange(n_dst) * ifx).astype(np.int64), n_src - 1)

    def one(n, p):
        sn = max(1, n // int(p))
        return np.ascontiguousarray(nn(sn, n)[nn(n, sn)].astype(np.int32))
    return one(w, pixel_size), one(h, pixel_size)


def resize_linear_axis(n_dst: int, n_src: int):
    """One axis of cv2.resize(..., INTER_LINEAR) on float data: for every destination index the source
    index of the first tap (int32) and the weight of the second tap (float32); the first tap's
    weight is 1 - w.  fx = (float)((dx + 0.5) * (n_src / n_dst) - 0.5); sx = floor(fx); fx -= sx;
    clamped to (0, 0) on the left and (n_src - 1, 0) on the right."""
    scale = n_src / n_dst
    fx = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int32)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo = sx < 0
    fx[lo] = 0.0
    sx[lo] = 0
    hi = sx >= n_src - 1
    fx[hi] = 0.0
    sx[hi] = n_src - 1
    return np.ascontiguousarray(sx), np.ascontiguousarray(fx)


def glitch_band(h2: int, glitch_height_frac: float):
    """ref:667 / :838 — first row of the bottom band."""
    return max(0, min(h2, h2 - int(h2 * glitch_height_frac)))


def glitch_offsets_render_segments(h2: int, w2: int, scanline_phase_px: float, glitch_amp_px: int, glitch_height_frac: float):
    """ref:838-855 — render-path glitch in the compact form the kernels take: (y0, int32 offsets (rows, num_segs),
    seg_len) or (y0, None, 0).  Pixel x of band row r is shifted by offsets[r, x // seg_len]: the reference expands
    `base[r] + seg_offsets[r, x // seg_len]` to every pixel before rounding (ref:852-855), which is the same value
    for all pixels of a segment — so the float32 add and np.rint are done once per (row, segment) here
    (4K: 311 KB per frame to upload instead of 10 MB)."""
    y0 = glitch_band(h2, glitch_height_frac)
    if y0 >= h2:
        return y0, None, 0
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
    return y0, np.ascontiguousarray(np.rint(base[:, None] + seg_offsets).astype(np.int32)), int(seg_len)


def glitch_offsets_render(h2: int, w2: int, scanline_phase_px: float, glitch_amp_px: int, glitch_height_frac: float):
    """The same offsets expanded to one per pixel, as ref:852-855 holds them: (y0, int32 (rows, w2)) or (y0, None)."""
    y0, seg, seg_len = glitch_offsets_render_segments(h2, w2, scanline_phase_px, glitch_amp_px, glitch_height_frac)
    if seg is None:
        return y0, None
    return y0, np.ascontiguousarray(seg[:, np.arange(w2) // seg_len])


def glitch_offsets_preview(h2: int, w2: int, scanline_phase_px: float, glitch_amp_px: int, glitch_height_frac: float):
    """ref:667-682 — preview-path glitch: (y0, int32 offsets (rows, 1)) or (y0, None)."""
    y0 = glitch_band(h2, glitch_height_frac)
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
    return y0, np.ascontiguousarray(np.rint(offs_row).astype(np.int32)[:, None])


def ptr(a) -> int:
    return 0 if a is None else a.ctypes.data
