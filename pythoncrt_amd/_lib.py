"""ctypes binding of libcrtfx.so (include/crtfx.h).  No torch types cross this boundary:
pointers are integers (tensor.data_ptr()), sizes are ints.  Missing library = hard error —
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("CRTFX_LIB") or os.path.join(_HERE, "libcrtfx.so")   # CRTFX_LIB: dev A/B builds only
CSRC = os.path.join(_HERE, "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("crtfx.hip", "crtfx_rr.hip", "crtfx_kernels.hip.h", "crtfx_common.hip.h", "crtfx_blur.hip.h", "crtfx_point.hip.h",
                                                "crtfx_phosphor.hip.h", "crtfx_phosphor_ct.hip.h", "crtfx_warp.hip.h", "crtfx_internal.h")] + \
          [os.path.join(ROOT, "include", "crtfx.h")]
RR_RADII = tuple(range(1, 31))      # one register-window build per radius up to 30 (crtfx_internal.h); larger radii: the split path

OK, E_INVALID, E_HIP, E_UNSUPPORTED, E_NOMEM = 0, -1, -2, -3, -4
PIX_U8, PIX_F16 = 0, 1
BLEND_NONE, BLEND_RENDER, BLEND_PREVIEW = 0, 1, 2

F_SATURATION, F_TEMPERATURE, F_BRIGHTCON, F_GAMMA = 1 << 0, 1 << 1, 1 << 2, 1 << 3
F_BLOOM, F_BLOOM_FAST, F_BLOOM_THR = 1 << 4, 1 << 5, 1 << 6
F_TRIAD, F_TRIAD_LUT, F_TRIAD_LUMA = 1 << 7, 1 << 8, 1 << 9
F_SCANLINES, F_VIGNETTE, F_FLICKER, F_NOISE, F_WARP, F_PIXELATE = 1 << 10, 1 << 11, 1 << 12, 1 << 13, 1 << 14, 1 << 15

_vp = ctypes.c_void_p


class CrtfxParams(ctypes.Structure):
    _fields_ = [
        ("size", ctypes.c_uint32), ("flags", ctypes.c_uint32),
        ("aberration_px", ctypes.c_int32), ("grain_size", ctypes.c_int32),
        ("bloom_radius", ctypes.c_int32), ("reserved0", ctypes.c_int32),
        ("saturation", ctypes.c_float), ("r_gain", ctypes.c_float), ("b_gain", ctypes.c_float),
        ("contrast", ctypes.c_float), ("brightness", ctypes.c_float), ("inv_gamma", ctypes.c_float),
        ("bloom_thr", ctypes.c_float), ("bloom_thr_den", ctypes.c_float), ("bloom_strength", ctypes.c_float),
        ("noise_scale", ctypes.c_float), ("warp_k", ctypes.c_float), ("warp_cx", ctypes.c_float), ("warp_cy", ctypes.c_float),
        ("vignette_strength", ctypes.c_double),
        ("bloom_taps", _vp), ("triad_row", _vp), ("lut_g", _vp), ("lut_inv", _vp),
        ("vig_nx2", _vp), ("vig_ny2", _vp), ("warp_xhat", _vp), ("warp_yhat", _vp),
        ("pix_xmap", _vp), ("pix_ymap", _vp),
        ("triad_full_dev", _vp), ("vignette_full_dev", _vp),
        ("grain_xofs", _vp), ("grain_xw", _vp), ("grain_yofs", _vp), ("grain_yw", _vp),
        ("grain_w", ctypes.c_int32), ("grain_h", ctypes.c_int32),
        ("fbu_xofs", _vp), ("fbu_xw", _vp), ("fbu_yofs", _vp), ("fbu_yw", _vp),
        ("fbd_xofs", _vp), ("fbd_xw", _vp), ("fbd_yofs", _vp), ("fbd_yw", _vp),
        ("grade_lut", _vp),
    ]


class CrtfxFrame(ctypes.Structure):
    _fields_ = [
        ("scan_row_dev", _vp), ("scan_plane_dev", _vp), ("noise_plane_dev", _vp),
        ("overlay_rgba_dev", _vp), ("glitch_offs_dev", _vp),
        ("flicker_factor", ctypes.c_double), ("noise_seed", ctypes.c_uint64), ("frame_index", ctypes.c_uint64),
        ("overlay_after", ctypes.c_int32), ("glitch_y0", ctypes.c_int32), ("glitch_cols", ctypes.c_int32),
        ("glitch_seg_len", ctypes.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/crtfx.h declares
SYMBOLS = {
    "crtfx_version": (ctypes.c_int, []),
    "crtfx_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "crtfx_destroy": (ctypes.c_int, [_vp]),
    "crtfx_last_error": (ctypes.c_char_p, [_vp]),
    "crtfx_set_params": (ctypes.c_int, [_vp, ctypes.POINTER(CrtfxParams)]),
    "crtfx_apply_static": (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(CrtfxFrame), _vp]),
    "crtfx_apply": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.POINTER(CrtfxFrame), _vp]),
    "crtfx_blend_quantise": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_double, _vp]),
    "crtfx_halo_correct_quantise": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _vp, _vp, _vp]),
    "crtfx_process_batch": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_int,
                                           ctypes.POINTER(CrtfxFrame), _vp, ctypes.c_double, ctypes.c_int, _vp, _vp]),
    "crtfx_noise_plane": (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp]),
    "crtfx_halo_correct_batch": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "crtfx_warp_map": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "crtfx_resize_state": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "crtfx_scanline_plane": (ctypes.c_int, [_vp] + [ctypes.c_double] * 5 + [_vp, _vp]),
    "crtfx_set_option": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "crtfx_debug_buffer": (ctypes.c_int, [_vp, _vp]),
    "crtfx_profile_enable": (ctypes.c_int, [_vp, ctypes.c_int]),
    "crtfx_profile_read": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "crtfx_last_plan": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "crtfx_kernel_lds_bytes": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "crtfx_host_blur_row": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int]),
}

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """Compile libcrtfx.so in-tree for gfx950 (hipcc cross-compiles without a GPU): crtfx.hip plus
    crtfx_rr.hip once per radius, the translation units in parallel, then one link."""
    out = out or LIB_PATH
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in SOURCES):
        return out
    from concurrent.futures import ThreadPoolExecutor
    hipcc = "hipcc" if _which("hipcc") else "/opt/rocm/bin/hipcc"
    objdir = os.path.join(ROOT, "build", "obj", os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    jobs = [([hipcc, *HIPCC_FLAGS, *extra_flags, *inc, "-c", os.path.join(CSRC, "crtfx.hip"), "-o", os.path.join(objdir, "crtfx.o")])]
    for r in RR_RADII:
        jobs.append([hipcc, *HIPCC_FLAGS, *extra_flags, *inc, f"-DRR_R={r}", "-c", os.path.join(CSRC, "crtfx_rr.hip"),
                     "-o", os.path.join(objdir, f"crtfx_rr_{r}.o")])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL if not verbose else None)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    objs = [j[-1] for j in jobs]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


def build_variant(name: str, extra_flags, radii=(9,), main_tu: bool = False) -> str:
    """Dev A/B builds (tools/ab.sh; never the product library): build/ab/libcrtfx_<name>.so = the in-tree library's objects with the
    translation units of `radii` (and crtfx.hip when main_tu) recompiled with extra_flags.  Load one with CRTFX_LIB=<path>."""
    build()
    hipcc = "hipcc" if _which("hipcc") else "/opt/rocm/bin/hipcc"
    base = os.path.join(ROOT, "build", "obj", os.path.basename(LIB_PATH))
    objdir = os.path.join(ROOT, "build", "obj", f"ab_{name}")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "build", "ab"), exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    objs = []
    for r in RR_RADII:
        o = os.path.join(base, f"crtfx_rr_{r}.o")
        if r in radii:
            o = os.path.join(objdir, f"crtfx_rr_{r}.o")
            subprocess.run([hipcc, *HIPCC_FLAGS, *extra_flags, *inc, f"-DRR_R={r}", "-c", os.path.join(CSRC, "crtfx_rr.hip"), "-o", o], check=True)
        objs.append(o)
    main = os.path.join(base, "crtfx.o")
    if main_tu:
        main = os.path.join(objdir, "crtfx.o")
        subprocess.run([hipcc, *HIPCC_FLAGS, *extra_flags, *inc, "-c", os.path.join(CSRC, "crtfx.hip"), "-o", main], check=True)
    out = os.path.join(ROOT, "build", "ab", f"libcrtfx_{name}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, main, *objs], check=True)
    return out


def _which(name):
    from shutil import which
    return which(name)


_LIB = None


def load() -> ctypes.CDLL:
    """dlopen the in-tree libcrtfx.so and type every entry point.  Raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or pythoncrt_amd._lib.build()). "
            "pythoncrt_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.crtfx_version() != 1:
        raise RuntimeError(f"libcrtfx ABI {lib.crtfx_version()} != 1")
    _LIB = lib
    return lib


class CrtfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcrtfx error {code}: {msg}")
        self.code = code


def check(lib, ctx, rc):
    if rc != OK:
        msg = lib.crtfx_last_error(ctx)
        raise CrtfxError(rc, msg.decode() if msg else "")
