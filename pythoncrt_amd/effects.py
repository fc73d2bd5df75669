"""Drop-in for the reference's per-frame API (crt_filter.py, `ref:LINE`):

    make_triad_mask(h, w, strength, softness_px=0.0)      ref:220
    make_vignette(h, w, strength)                         ref:266
    apply_crt_effect(frame, ...) -> (out_u8, img_float)   ref:531-699
    apply_static_effects(frame, ...) -> img_float         ref:702-861

Same names, argument order, defaults and gating as the reference.  Frames may be numpy
`uint8` H x W x 3 arrays (results come back as numpy arrays) or `torch.uint8` tensors on a ROCm
device (results stay on that device).  The work is done by libcrtfx.so (hand-written gfx950
kernels) through ctypes; if the library or a GPU is missing these functions raise — there is no
CPU path here.

Differences from the reference, by design:
  * the mask builders return light descriptors (`TriadMask`, `VignetteMask`) instead of
    H x W x 3 float32 / H x W float64 arrays (99.5 MB + 66 MB at 4K): the kernels evaluate the
    masks from one row / two axis vectors.  `np.asarray(mask)` still yields the reference's array,
    and plain arrays are accepted too.
  * float results are float32 (the reference's image becomes float64 after the vignette /
    flicker multiply; the kernels do that tail in float64 and narrow once at the store).
  * grain comes from a counter-based RNG keyed by (noise_seed, frame_index) — cv2.randn in the
    reference is unseeded and thread-local, so its grain is unreproducible by construction.
    Extra keyword-only arguments `noise_seed`, `frame_index`, `noise_plane` control it.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib, tables

# ---------------------------------------------------------------------------------------
# mask descriptors
# ---------------------------------------------------------------------------------------


class TriadMask:
    """What make_triad_mask returns: the (w, 3) row the reference repeats h times (ref:230)."""

    def __init__(self, h: int, w: int, strength: float, softness_px: float, row: np.ndarray):
        self.h, self.w, self.strength, self.softness_px = int(h), int(w), float(strength), float(softness_px)
        self.row = row
        self.shape = (self.h, self.w, 3)
        self.dtype = np.dtype(np.float32)
        self.ndim = 3

    def __array__(self, dtype=None, copy=None):
        full = np.repeat(self.row[None, :, :], self.h, axis=0)
        return full if dtype is None else full.astype(dtype)

    def key(self):
        if self.strength != self.strength:       # recognised from a plain array: identified by the row itself
            return ("triad_row", self.h, self.w, hash(self.row.tobytes()))
        return ("triad_row", self.h, self.w, self.strength, self.softness_px)


class VignetteMask:
    """What make_vignette returns: v = 1 - strength * clip(nx^2 + ny^2, 0, 1) (ref:266-276)."""

    def __init__(self, h: int, w: int, strength: float):
        self.h, self.w, self.strength = int(h), int(w), float(strength)
        self.shape = (self.h, self.w)
        self.dtype = np.dtype(np.float64)
        self.ndim = 2

    def __array__(self, dtype=None, copy=None):
        full = tables.vignette_full(self.h, self.w, self.strength)
        return full if dtype is None else full.astype(dtype)

    def key(self):
        return ("vig_axes", self.h, self.w, self.strength)


# A caller that rebinds only apply_* keeps building its masks with the reference's own make_triad_mask /
# make_vignette and hands over H x W x 3 float32 / H x W float64 numpy arrays (99.5 MB + 66 MB at 4K).  Those have the
# structure the descriptors encode — identical rows; 1 - s * clip(nx^2 + ny^2) — so they are recognised once per
# array object (verified element for element, remembered by identity) and take the same fast path; anything else
# stays a per-pixel plane.
# The verdict is remembered per array OBJECT: the entry keeps a reference to the array (so its id cannot be reused by
# another array while the entry lives) together with a fingerprint of a strided sample (an in-place rebuild of the
# mask is then looked at again).  At most four arrays are held.
_RECOGNISED = {}      # (kind, id(array)) -> (array, fingerprint, descriptor or False)
_RECOGNISED_MAX = 4
_RECOGNISED_LOCK = threading.Lock()     # apply_static_effects is called from the reference's two worker threads


def _fingerprint(a: np.ndarray) -> int:
    step = max(1, a.shape[0] // 8)
    return hash((a.shape, a[::step, :: max(1, a.shape[1] // 64)].tobytes()))


def _lookup(kind, mask):
    with _RECOGNISED_LOCK:
        e = _RECOGNISED.get((kind, id(mask)))
    if e is not None and e[0] is mask and e[1] == _fingerprint(mask):
        return e[2]
    return None


def _remember(kind, mask, value):
    fp = _fingerprint(mask)
    with _RECOGNISED_LOCK:
        _RECOGNISED.pop((kind, id(mask)), None)
        while len(_RECOGNISED) >= _RECOGNISED_MAX:
            _RECOGNISED.pop(next(iter(_RECOGNISED)), None)
        _RECOGNISED[(kind, id(mask))] = (mask, fp, value)
    return value


def _recognise_triad(mask):
    if not isinstance(mask, np.ndarray) or mask.ndim != 3 or mask.shape[2] != 3 or mask.dtype != np.float32 or mask.shape[0] < 1:
        return mask
    hit = _lookup("triad", mask)
    if hit is None:
        row = np.ascontiguousarray(mask[0])
        same = bool((mask[:: max(1, mask.shape[0] // 16)] == row).all()) and bool((mask == row).all())      # cheap sample first
        hit = _remember("triad", mask, TriadMask(mask.shape[0], mask.shape[1], float("nan"), float("nan"), row.copy()) if same else False)
    return hit if hit else mask


def _recognise_vignette(mask):
    if not isinstance(mask, np.ndarray) or mask.ndim != 2 or mask.dtype != np.float64 or mask.size == 0:
        return mask
    hit = _lookup("vig", mask)
    if hit is None:
        h, w = mask.shape
        nx2, ny2 = tables.vignette_axes(h, w)
        found = False
        if nx2[0] + ny2[0] >= 1.0:                       # the corner sits on the clip: v = 1 - s there
            s0 = 1.0 - float(mask[0, 0])
            cands = {s0, round(s0, 6), round(s0, 4), round(s0, 3), round(s0, 2)}
            for k in range(1, 4):
                cands.add(float(np.nextafter(s0, 2.0)) if k == 1 else float(np.nextafter(s0, -1.0)) if k == 2 else s0)
            ys, xs = np.arange(0, h, max(1, h // 32)), np.arange(0, w, max(1, w // 32))
            sample = mask[np.ix_(ys, xs)]
            for s in sorted(cands):
                if 0.0 <= s <= 1.0 and np.array_equal(sample, 1.0 - s * np.clip(nx2[None, xs] + ny2[ys, None], 0.0, 1.0)) \
                        and np.array_equal(mask, tables.vignette_full(h, w, s)):
                    found = VignetteMask(h, w, s)
                    break
        hit = _remember("vig", mask, found)
    return hit if hit else mask


class DeviceState:
    """The float image `apply_crt_effect` returns for a numpy frame (its second result, ref:699): an array-like that OWNS
    the device tensor and materialises a numpy array only when one is asked for (`np.asarray`, indexing, arithmetic, any
    ndarray attribute).  The reference's GUI tick only threads the value back in —
    `out, self.prev_img = apply_crt_effect(..., state_prev=self.prev_img)` (ref:1810-1852) — and handed back as
    `state_prev` it is used where it lies: no 25 MB download and upload per 1080p tick.  `np.asarray(state)` is the
    float32 array the call returned before."""

    __array_priority__ = 100.0

    def __init__(self, tensor: torch.Tensor):
        self._t = tensor
        self._host = None
        self._dirty = False                     # the host copy was written to: it is the truth until uploaded again
        self.shape = tuple(tensor.shape)
        self.dtype = np.dtype(np.float32)
        self.ndim = tensor.ndim
        self.size = int(tensor.numel())

    @property
    def tensor(self) -> torch.Tensor:
        """The device tensor (float32 H x W x 3); re-uploaded first if the host copy was modified through this object."""
        if self._dirty:
            self._t = torch.from_numpy(np.array(self._host, dtype=np.float32, order="C")).to(self._t.device)
            self._dirty = False
        return self._t

    def _host_rw(self) -> np.ndarray:
        if self._host is None:
            self._host = self._t.cpu().numpy()
        return self._host

    def numpy(self) -> np.ndarray:
        """The float32 host array, READ-ONLY: every array this object hands out (np.asarray, indexing, ndarray attributes) is a
        non-writeable view of its cached host copy, so a write the object cannot see — `np.asarray(state)[...] = x`,
        `state.fill(0)`, `np.clip(..., out=np.asarray(state))` — raises instead of silently leaving the device tensor stale.
        Write through the object itself (`state[k] = v`: uploaded again before the next tick) or take a copy (`np.array(state)`)
        and pass that back as `state_prev`."""
        v = self._host_rw().view()
        v.flags.writeable = False
        return v

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, k):
        return self.numpy()[k]

    def __setitem__(self, k, v):
        self._host_rw()[k] = v
        self._dirty = True

    def __getattr__(self, name):                # anything else an ndarray has (astype, mean, min, T, tobytes, ...)
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    def __repr__(self):
        return f"DeviceState(shape={self.shape}, float32, device={self._t.device}, {'materialised' if self._host is not None else 'device-resident'})"


def _state_op(name):
    def op(self, *args):
        return getattr(self.numpy(), name)(*[a.numpy() if isinstance(a, DeviceState) else a for a in args])
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "floordiv", "pow", "mod", "radd", "rsub", "rmul", "rtruediv", "rfloordiv", "rpow", "rmod",
           "neg", "pos", "abs", "lt", "le", "gt", "ge", "eq", "ne", "matmul", "rmatmul"):
    setattr(DeviceState, f"__{_n}__", _state_op(f"__{_n}__"))
DeviceState.__hash__ = None


def make_triad_mask(h: int, w: int, strength: float, softness_px: float = 0.0) -> TriadMask:
    """ref:220-235."""
    return TriadMask(h, w, strength, softness_px, tables.triad_row(_lib.load(), int(w), strength, softness_px))


def make_vignette(h: int, w: int, strength: float) -> VignetteMask:
    """ref:266-276."""
    return VignetteMask(h, w, strength)


# ---------------------------------------------------------------------------------------
# engine: one libcrtfx ctx per (device, H, W, thread)
# ---------------------------------------------------------------------------------------

_tls = threading.local()
_ENGINES_PER_THREAD = 6     # contexts kept per thread (one per frame size / pixel format in use)
_seed_lock = threading.Lock()
_seed_counter = [int.from_bytes(os.urandom(8), "little")]


def _fresh_seed() -> int:
    with _seed_lock:
        _seed_counter[0] = (_seed_counter[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return _seed_counter[0]


# Testing / tuning switches handed to every ctx created from now on (crtfx_set_option; include/crtfx.h): e.g.
# {"FORCE_GENERIC": 1}.  Empty in the product path; nothing reads the environment.
DEBUG_OPTIONS = {}
_OPTION_IDS = {"FORCE_GENERIC": 1, "FORCE_RUNTIME_FLAGS": 2, "NO_CC": 3, "GROUP": 4, "SEG_ROWS": 5, "WARP_ROWS": 6, "POINT_TILES": 7,
               "OVERLAP": 8, "DEBUG_PLAN": 9, "FORCE_CC": 10, "SPLIT_FROM": 11, "SPLIT_SRC_PLANE": 12, "NO_CT": 13, "NO_PLAIN_WARP": 14, "BAND_MB": 15,
               "NO_FUSED_HALF": 16}


class Engine:
    """Owns a crtfx ctx, the host tables and the device-side per-frame scratch tensors."""

    def __init__(self, device: torch.device, h: int, w: int, pix_fmt: int = _lib.PIX_U8):
        self.pix_fmt = pix_fmt
        if device.type != "cuda":
            raise RuntimeError(f"pythoncrt_amd needs a ROCm device, got {device}")
        self.lib = _lib.load()
        self.device, self.h, self.w = device, int(h), int(w)
        ctx = ctypes.c_void_p()
        rc = self.lib.crtfx_create(device.index if device.index is not None else torch.cuda.current_device(),
                                   self.h, self.w, pix_fmt, ctypes.byref(ctx))
        if rc != _lib.OK:
            raise _lib.CrtfxError(rc, f"crtfx_create({device}, {h}, {w}) failed")
        self.ctx = ctx
        for name, value in DEBUG_OPTIONS.items():
            _lib.check(self.lib, ctx, self.lib.crtfx_set_option(ctx, _OPTION_IDS[name], int(value)))
        self.params_key = None
        self.keep = {}            # host arrays / device tensors the current params point at
        self.auto_frame = 0
        self.seed = _fresh_seed()

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.crtfx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def last_plan(self) -> dict:
        """crtfx_last_plan as a dict: which build of each kernel class the last apply / process_batch call launched."""
        buf = ctypes.create_string_buffer(512)
        _lib.check(self.lib, self.ctx, self.lib.crtfx_last_plan(self.ctx, buf, len(buf)))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                k, _, v = item.partition("=")
                out[k] = int(v) if v.lstrip("-").isdigit() else v
        return out

    # -- params ------------------------------------------------------------------------
    def set_params(self, s: "Settings"):
        key = s.key()
        if key == self.params_key:
            return
        h, w = self.h, self.w
        p = _lib.CrtfxParams()
        p.size = ctypes.sizeof(_lib.CrtfxParams)
        keep = {}
        flags = 0
        p.aberration_px = int(s.aberration_px)
        p.grain_size = int(s.grain_size) if s.grain_size else 1
        # colour grade gates: exact comparisons as in ref:288-302
        if s.saturation != 1.0:
            flags |= _lib.F_SATURATION
        if s.temperature != 0.0:
            flags |= _lib.F_TEMPERATURE
            t = float(s.temperature)
            p.r_gain = float(np.clip(1.0 + 0.5 * t, 0.5, 1.5))
            p.b_gain = float(np.clip(1.0 - 0.5 * t, 0.5, 1.5))
        if s.brightness != 0.0 or s.contrast != 1.0:
            flags |= _lib.F_BRIGHTCON
        if s.gamma != 1.0 and s.gamma > 0.0:
            flags |= _lib.F_GAMMA
            p.inv_gamma = 1.0 / float(s.gamma)
        p.saturation, p.contrast, p.brightness = float(s.saturation), float(s.contrast), float(s.brightness)
        if self.pix_fmt == _lib.PIX_U8 and not (flags & _lib.F_SATURATION) and (flags & (_lib.F_TEMPERATURE | _lib.F_BRIGHTCON | _lib.F_GAMMA)):
            keep["grade_lut"] = tables.grade_lut(s.brightness, s.contrast, s.gamma, s.temperature)
            p.grade_lut = tables.ptr(keep["grade_lut"])
        if s.pixel_size > 1:
            flags |= _lib.F_PIXELATE
            keep["xmap"], keep["ymap"] = tables.pixelate_maps(h, w, s.pixel_size)
            p.pix_xmap, p.pix_ymap = tables.ptr(keep["xmap"]), tables.ptr(keep["ymap"])
        # bloom gate ref:599
        if s.bloom_strength > 0.0 and (s.bloom_sigma > 0.0 or s.fast_bloom):
            flags |= _lib.F_BLOOM
            if s.fast_bloom:
                flags |= _lib.F_BLOOM_FAST
                hw, hh = max(1, w // 2), max(1, h // 2)                                  # ref:606
                keep["fbu"] = tables.resize_linear_axis(w, hw) + tables.resize_linear_axis(h, hh)
                p.fbu_xofs, p.fbu_xw, p.fbu_yofs, p.fbu_yw = (tables.ptr(a) for a in keep["fbu"])
                if not (hw * 2 == w and hh * 2 == h):      # not OpenCV's exact-2x INTER_AREA shortcut
                    keep["fbd"] = tables.resize_linear_axis(hw, w) + tables.resize_linear_axis(hh, h)
                    p.fbd_xofs, p.fbd_xw, p.fbd_yofs, p.fbd_yw = (tables.ptr(a) for a in keep["fbd"])
            else:
                k = tables.bloom_ksize(s.bloom_sigma)
                keep["taps"] = tables.gaussian_taps(k, s.bloom_sigma) if k > 1 else np.ones(1, np.float32)
                p.bloom_radius = (k - 1) // 2
                p.bloom_taps = tables.ptr(keep["taps"])
            if s.bloom_threshold > 0.0:
                flags |= _lib.F_BLOOM_THR
                thr = float(min(0.99, max(0.0, s.bloom_threshold)))
                p.bloom_thr, p.bloom_thr_den = thr, max(1e-6, 1.0 - thr)
            p.bloom_strength = float(s.bloom_strength)
        # triad ref:613
        if s.triad_mask is not None:
            flags |= _lib.F_TRIAD
            tm = s.triad_mask
            if isinstance(tm, TriadMask):
                _check_shape(tm.shape, (h, w, 3), "triad_mask")
                keep["triad_row"] = np.ascontiguousarray(tm.row, np.float32)
                p.triad_row = tables.ptr(keep["triad_row"])
            else:
                full = _as_device_tensor(tm, self.device, torch.float32, (h, w, 3), "triad_mask")
                keep["triad_full"] = full
                p.triad_full_dev = full.data_ptr()
            if tables.triad_uses_lut(s.triad_gamma, s.triad_preserve_luma):
                flags |= _lib.F_TRIAD_LUT
                if s.triad_preserve_luma:
                    flags |= _lib.F_TRIAD_LUMA
                keep["lut_g"], keep["lut_inv"] = tables.triad_luts(s.triad_gamma)
                p.lut_g, p.lut_inv = tables.ptr(keep["lut_g"]), tables.ptr(keep["lut_inv"])
        if s.scanline_strength > 0.0:
            flags |= _lib.F_SCANLINES
        if s.vignette_mask is not None:
            flags |= _lib.F_VIGNETTE
            vm = s.vignette_mask
            if isinstance(vm, VignetteMask):
                _check_shape(vm.shape, (h, w), "vignette_mask")
                keep["nx2"], keep["ny2"] = tables.vignette_axes(h, w)
                p.vig_nx2, p.vig_ny2 = tables.ptr(keep["nx2"]), tables.ptr(keep["ny2"])
                p.vignette_strength = vm.strength
            else:
                full = _as_device_tensor(vm, self.device, torch.float64, (h, w), "vignette_mask")
                keep["vig_full"] = full
                p.vignette_full_dev = full.data_ptr()
        if s.flicker_strength > 0.0 and s.flicker_hz > 0.0:
            flags |= _lib.F_FLICKER
        if s.noise_strength > 0.0:
            flags |= _lib.F_NOISE
            p.noise_scale = s.noise_strength / 255.0
            if s.grain_size and s.grain_size > 1:                                       # ref:637-642
                gh, gw = max(1, h // int(s.grain_size)), max(1, w // int(s.grain_size))
                keep["grain"] = tables.resize_linear_axis(w, gw) + tables.resize_linear_axis(h, gh)
                p.grain_xofs, p.grain_xw, p.grain_yofs, p.grain_yw = (tables.ptr(a) for a in keep["grain"])
                p.grain_w, p.grain_h = gw, gh
        if s.warp_strength != 0.0:
            flags |= _lib.F_WARP
            keep["xhat"], keep["yhat"], cx, cy = tables.warp_axes(h, w)
            p.warp_xhat, p.warp_yhat = tables.ptr(keep["xhat"]), tables.ptr(keep["yhat"])
            p.warp_k, p.warp_cx, p.warp_cy = float(s.warp_strength) * 0.5, cx, cy
        p.flags = flags
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.ctx, self.lib.crtfx_set_params(self.ctx, ctypes.byref(p)))
        self.keep = keep
        self.flags = flags
        self.params_key = key

    # -- per-frame record ----------------------------------------------------------------
    def frame_record(self, s: "Settings", scanline_phase_px: float, time_sec: float, noise_seed, frame_index, noise_plane, hold: list,
                     overlay=None, overlay_after: bool = True, glitch=None):
        f = _lib.CrtfxFrame()
        h, w = self.h, self.w
        if overlay is not None:
            ov = _overlay_tensor(overlay, self.device, h, w)
            hold.append(ov)
            f.overlay_rgba_dev = ov.data_ptr()
            f.overlay_after = 1 if overlay_after else 0
        if glitch is not None and glitch[1] is not None:
            y0, offs = glitch[0], glitch[1]
            t = torch.from_numpy(offs).to(self.device)
            hold.append(t)
            f.glitch_offs_dev, f.glitch_y0, f.glitch_cols = t.data_ptr(), int(y0), int(offs.shape[1])
            f.glitch_seg_len = int(glitch[2]) if len(glitch) > 2 else 0
        if s.scanline_strength > 0.0:
            if s.scanline_angle == 0.0 and s.scanline_thickness == 1.0:       # ref:619
                row = tables.scanline_rows(h, s.scanline_strength, s.scanline_period_px, [scanline_phase_px])[0]
                t = torch.from_numpy(row).to(self.device, non_blocking=False)
                f.scan_row_dev = t.data_ptr()
            else:
                # make_scanline_mask_2d (ref:308-328) on the device: the host version is an H x W float64 sin/pow per
                # call (40 ms at 1080p, a stall per GUI tick); within one float32 ulp of it (crtfx_scanline_plane)
                t = torch.empty((h, w), dtype=torch.float32, device=self.device)
                omega, tan_t, inv_sharp = tables.scanline_plane_scalars(s.scanline_period_px, s.scanline_angle, s.scanline_thickness)
                with torch.cuda.device(self.device):
                    _lib.check(self.lib, self.ctx, self.lib.crtfx_scanline_plane(
                        self.ctx, float(s.scanline_strength), omega, float(scanline_phase_px), tan_t, inv_sharp, t.data_ptr(),
                        _stream_ptr(self.device)))
                f.scan_plane_dev = t.data_ptr()
            hold.append(t)
        f.flicker_factor = tables.flicker_factor(s.flicker_strength, s.flicker_hz, time_sec) if (self.flags & _lib.F_FLICKER) else 1.0
        if noise_plane is not None:
            t = _as_device_tensor(noise_plane, self.device, torch.float32, None, "noise_plane")
            hold.append(t)
            f.noise_plane_dev = t.data_ptr()
        f.noise_seed = (self.seed if noise_seed is None else int(noise_seed)) & 0xFFFFFFFFFFFFFFFF
        if frame_index is None:
            frame_index = self.auto_frame
            self.auto_frame += 1
        f.frame_index = int(frame_index) & 0xFFFFFFFFFFFFFFFF
        return f


def _overlay_tensor(ov, device, h, w) -> torch.Tensor:
    """ref:590-594 — non-uint8 overlays are clipped to 0..255 and cast; an overlay of another size goes
    through Pillow's bilinear resize on the host exactly as in the reference (text.fit_overlay)."""
    if isinstance(ov, torch.Tensor):
        t = ov if ov.dtype == torch.uint8 else ov.clamp(0, 255).to(torch.uint8)
    else:
        a = np.asarray(ov)
        if a.dtype != np.uint8:
            a = np.clip(a, 0, 255).astype(np.uint8)
        t = torch.from_numpy(np.array(a, order="C"))      # own copy: the caller's array may be read-only
    if t.ndim != 3 or t.shape[2] != 4:
        raise ValueError(f"text_overlay_rgba must be H x W x 4, got {tuple(t.shape)}")
    if t.shape[0] != h or t.shape[1] != w:
        from .text import fit_overlay
        t = torch.from_numpy(np.array(fit_overlay(t.cpu().numpy(), h, w), order="C"))
    return t.to(device).contiguous()


def _check_shape(got, want, name):
    if tuple(got) != tuple(want):
        raise ValueError(f"{name} has shape {tuple(got)}, frame needs {tuple(want)}")


def _as_device_tensor(a, device, dtype, shape, name) -> torch.Tensor:
    if isinstance(a, DeviceState):              # a state this module returned: already on the device
        a = a.tensor
    if isinstance(a, torch.Tensor):
        t = a.to(device=device, dtype=dtype).contiguous()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(device=device, dtype=dtype).contiguous()
    if shape is not None:
        _check_shape(t.shape, shape, name)
    return t


def _engine(device: torch.device, h: int, w: int, pix_fmt: int = _lib.PIX_U8) -> Engine:
    cache = getattr(_tls, "engines", None)
    if cache is None:
        cache = _tls.engines = {}
    k = (device.index, h, w, pix_fmt)
    e = cache.pop(k, None)
    if e is None:
        e = Engine(device, h, w, pix_fmt)
    cache[k] = e                      # most recently used last
    while len(cache) > _ENGINES_PER_THREAD:      # a preview window dragged through many sizes must not pin a ctx
        cache.pop(next(iter(cache)))             # (scratch images, tables) per size for ever: Engine.__del__ frees it
    return e


class Settings:
    """The static keyword set shared by apply_crt_effect / apply_static_effects."""
    FIELDS = ("scanline_strength", "triad_mask", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma",
              "bloom_strength", "bloom_threshold", "noise_strength", "vignette_mask", "scanline_period_px", "fast_bloom",
              "pixel_size", "brightness", "contrast", "gamma", "saturation", "temperature", "flicker_strength",
              "flicker_hz", "grain_size", "scanline_angle", "scanline_thickness", "warp_strength")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    def key(self):
        out = []
        for f in self.FIELDS:
            v = getattr(self, f)
            if f in ("triad_mask", "vignette_mask") and v is not None:
                # a plain array / tensor mask is a per-pixel plane: keyed by identity AND a strided-sample fingerprint (or
                # the tensor's version counter), so a mask rebuilt in place — or an id reused after a free — is uploaded again
                if hasattr(v, "key"):
                    v = v.key()
                elif isinstance(v, torch.Tensor):
                    v = ("tensor", id(v), v.data_ptr(), tuple(v.shape), v._version)
                else:
                    v = ("array", id(v), _fingerprint(np.asarray(v)) if np.asarray(v).ndim >= 2 else hash(np.asarray(v).tobytes()))
            out.append(v)
        return tuple(out)


def _frame_to_device(frame):
    """-> (uint8 or float16 tensor H x W x 3 on a ROCm device, was_numpy).  A float16 frame is the
    reference's frame array held as half (same 0..255 scale, ref:569); its output frame is half too."""
    if isinstance(frame, torch.Tensor):
        if frame.dtype not in (torch.uint8, torch.float16) or frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError(f"frame tensor must be uint8 or float16 H x W x 3, got {frame.dtype} {tuple(frame.shape)}")
        if not frame.is_cuda:
            raise RuntimeError("frame tensor must live on a ROCm device (pass a numpy array for host frames)")
        return frame.contiguous(), False
    a = np.asarray(frame)
    if a.dtype not in (np.uint8, np.float16) or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"frame must be uint8 or float16 H x W x 3, got {a.dtype} {a.shape}")
    if not torch.cuda.is_available():
        raise RuntimeError("pythoncrt_amd: no ROCm device visible; there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    return _upload(a, dev), True


def _upload(a: np.ndarray, dev) -> torch.Tensor:
    """Host frame -> device through a pinned staging block (torch's caching host allocator hands the same few blocks round and
    keeps one alive until the copy that reads it has run): one memcpy + an asynchronous DMA instead of the driver's chunked
    staging of a pageable source, which also blocks the calling thread."""
    stage = torch.empty(a.shape, dtype=torch.uint8 if a.dtype == np.uint8 else torch.float16, pin_memory=True)
    np.copyto(stage.numpy(), a)
    return stage.to(dev, non_blocking=True)


def _download(t: torch.Tensor) -> np.ndarray:
    """Device tensor -> a fresh numpy array (backed by a pinned block that lives as long as the array does)."""
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def apply_crt_effect(
    frame,
    scanline_strength: float,
    triad_mask,
    triad_gamma: float,
    triad_preserve_luma: bool,
    aberration_px: int,
    bloom_sigma: float,
    bloom_strength: float,
    bloom_threshold: float,
    noise_strength: float,
    vignette_mask,
    persistence: float,
    state_prev,
    scanline_period_px: float,
    scanline_phase_px: float,
    fast_bloom: bool,
    pixel_size: int,
    glitch_amp_px: int = 0,
    glitch_height_frac: float = 0.0,
    time_sec: float = 0.0,
    brightness: float = 0.0,
    contrast: float = 1.0,
    gamma: float = 1.0,
    saturation: float = 1.0,
    temperature: float = 0.0,
    flicker_strength: float = 0.0,
    flicker_hz: float = 0.0,
    grain_size: int = 1,
    scanline_angle: float = 0.0,
    scanline_thickness: float = 1.0,
    warp_strength: float = 0.0,
    text_overlay_rgba=None,
    text_overlay_after: bool = True,
    *,
    noise_seed: Optional[int] = None,
    frame_index: Optional[int] = None,
    noise_plane=None,
) -> Tuple[object, object]:
    """ref:531-699 — full chain, preview-path persistence (cv2.addWeighted, ref:693) and
    quantise.  Returns (out_u8, img_float); img_float is the next call's `state_prev`.  For a numpy frame img_float is
    a `DeviceState`: float32 H x W x 3 to numpy (`np.asarray`, indexing, arithmetic), resident on the device until asked."""
    fr, was_numpy = _frame_to_device(frame)
    h, w = fr.shape[0], fr.shape[1]
    eng = _engine(fr.device, h, w, _lib.PIX_F16 if fr.dtype == torch.float16 else _lib.PIX_U8)
    triad_mask, vignette_mask = _recognise_triad(triad_mask), _recognise_vignette(vignette_mask)
    s = Settings(**{k: v for k, v in locals().items() if k in Settings.FIELDS})
    eng.set_params(s)
    hold = []
    glitch = None
    if glitch_amp_px > 0 and glitch_height_frac > 0.0:                      # ref:664 — preview variant
        glitch = tables.glitch_offsets_preview(h, w, scanline_phase_px, glitch_amp_px, glitch_height_frac)
    rec = eng.frame_record(s, scanline_phase_px, time_sec, noise_seed, frame_index, noise_plane, hold,
                           overlay=text_overlay_rgba, overlay_after=text_overlay_after, glitch=glitch)
    out = torch.empty((h, w, 3), dtype=fr.dtype, device=fr.device)
    blend = _lib.BLEND_NONE
    if state_prev is not None and persistence > 0.0:                       # ref:687
        prev = _as_device_tensor(state_prev, fr.device, torch.float32, None, "state_prev")
        if prev.ndim != 3 or prev.shape[2] != 3:
            raise ValueError(f"state_prev must be H x W x 3, got {tuple(prev.shape)}")
        if tuple(prev.shape) != (h, w, 3):                                 # ref:689-690: cv2.resize(state_prev, (w, h), INTER_LINEAR)
            state = torch.empty((h, w, 3), dtype=torch.float32, device=fr.device)
            with torch.cuda.device(fr.device):
                rc = eng.lib.crtfx_resize_state(eng.ctx, prev.data_ptr(), int(prev.shape[0]), int(prev.shape[1]), state.data_ptr(),
                                                _stream_ptr(fr.device))
            _lib.check(eng.lib, eng.ctx, rc)
        else:
            state = prev.clone()      # the reference never mutates state_prev
        blend = _lib.BLEND_PREVIEW
    else:
        state = torch.empty((h, w, 3), dtype=torch.float32, device=fr.device)
    with torch.cuda.device(fr.device):
        rc = eng.lib.crtfx_apply(eng.ctx, fr.data_ptr(), out.data_ptr(), state.data_ptr(), None, blend,
                                 float(persistence), ctypes.byref(rec), _stream_ptr(fr.device))
    _lib.check(eng.lib, eng.ctx, rc)
    # `hold` (per-frame tables) may be released here: the work above is enqueued on torch's current
    # stream and torch's caching allocator only reuses a freed block for later work on that stream.
    if was_numpy:
        # the frame goes back as a numpy array (ref:1853 wraps it in a QImage); the float state stays on the device behind
        # an array-like that materialises on demand and is taken back as `state_prev` without a copy (DeviceState)
        return _download(out), DeviceState(state)
    return out, state


def apply_static_effects(
    frame,
    scanline_strength: float,
    triad_mask,
    triad_gamma: float,
    triad_preserve_luma: bool,
    aberration_px: int,
    bloom_sigma: float,
    bloom_strength: float,
    bloom_threshold: float,
    noise_strength: float,
    vignette_mask,
    scanline_period_px: float,
    scanline_phase_px: float,
    fast_bloom: bool,
    pixel_size: int,
    glitch_amp_px: int,
    glitch_height_frac: float,
    time_sec: float = 0.0,
    brightness: float = 0.0,
    contrast: float = 1.0,
    gamma: float = 1.0,
    saturation: float = 1.0,
    temperature: float = 0.0,
    flicker_strength: float = 0.0,
    flicker_hz: float = 0.0,
    grain_size: int = 1,
    scanline_angle: float = 0.0,
    scanline_thickness: float = 1.0,
    warp_strength: float = 0.0,
    text_overlay_rgba=None,
    text_overlay_after: bool = True,
    *,
    noise_seed: Optional[int] = None,
    frame_index: Optional[int] = None,
    noise_plane=None,
):
    """ref:702-861 — stateless chain; returns the float image (float32 here)."""
    fr, was_numpy = _frame_to_device(frame)
    h, w = fr.shape[0], fr.shape[1]
    eng = _engine(fr.device, h, w, _lib.PIX_F16 if fr.dtype == torch.float16 else _lib.PIX_U8)
    triad_mask, vignette_mask = _recognise_triad(triad_mask), _recognise_vignette(vignette_mask)
    s = Settings(**{k: v for k, v in locals().items() if k in Settings.FIELDS})
    eng.set_params(s)
    hold = []
    glitch = None
    if glitch_amp_px > 0 and glitch_height_frac > 0.0:                      # ref:835 — render variant
        glitch = tables.glitch_offsets_render_segments(h, w, scanline_phase_px, glitch_amp_px, glitch_height_frac)
    rec = eng.frame_record(s, scanline_phase_px, time_sec, noise_seed, frame_index, noise_plane, hold,
                           overlay=text_overlay_rgba, overlay_after=text_overlay_after, glitch=glitch)
    img = torch.empty((h, w, 3), dtype=torch.float32, device=fr.device)
    with torch.cuda.device(fr.device):
        rc = eng.lib.crtfx_apply_static(eng.ctx, fr.data_ptr(), img.data_ptr(), ctypes.byref(rec), _stream_ptr(fr.device))
    _lib.check(eng.lib, eng.ctx, rc)
    return _download(img) if was_numpy else img
