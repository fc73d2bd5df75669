"""Host logic of the product (pythoncrt_amd/tables.py, effects.py descriptors) against the
oracle, plus the C-ABI surface: the library loads and exports every symbol include/crtfx.h
declares.  No GPU compute here."""
import os
import re

import numpy as np
import pytest

from oracle import crt_oracle as orc
from pythoncrt_amd import _lib, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def test_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "crtfx.h")).read()
    declared = set(re.findall(r"\b(crtfx_[a-z_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.crtfx_version() == 1


def test_option_ids_match_the_header():
    """effects._OPTION_IDS (the names DEBUG_OPTIONS / bench.py --opt take) == the crtfx_option enum of include/crtfx.h."""
    from pythoncrt_amd import effects
    hdr = open(os.path.join(ROOT, "include", "crtfx.h")).read()
    enum = {k: int(v) for k, v in re.findall(r"CRTFX_OPT_([A-Z_]+)\s*=\s*(\d+)", hdr)}
    assert enum and enum == effects._OPTION_IDS, set(enum.items()) ^ set(effects._OPTION_IDS.items())


def test_struct_layout_matches_header(lib):
    """crtfx_set_params rejects a struct whose size field disagrees with the C sizeof."""
    import ctypes
    p = _lib.CrtfxParams()
    p.size = ctypes.sizeof(_lib.CrtfxParams) - 4
    assert lib.crtfx_set_params(None, ctypes.byref(p)) == _lib.E_INVALID
    assert ctypes.sizeof(_lib.CrtfxParams) == 296 and ctypes.sizeof(_lib.CrtfxFrame) == 80   # == sizeof in C (gcc on include/crtfx.h)


def test_create_without_gpu_fails_cleanly(lib):
    import ctypes
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    assert lib.crtfx_create(0, 48, 64, 0, ctypes.byref(ctx)) < 0 and not ctx.value


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pythoncrt_amd
    frame = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pythoncrt_amd.apply_static_effects(frame, 0.6, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0)


def test_gaussian_taps_and_ksize():
    for s in (0.1, 0.17, 0.5, 0.83, 0.84, 1.2, 1.5, 2.5, 3.0, 3.5, 10.0):
        assert tables.bloom_ksize(s) == orc.bloom_ksize(s)
        k = tables.bloom_ksize(s)
        if k > 1:
            assert np.array_equal(tables.gaussian_taps(k, s), orc.gaussian_kernel(k, s))
    for s in (0.2, 0.5, 0.83, 0.84, 1.0, 1.5, 2.5, 4.0):
        assert tables.triad_ksize(s) == orc.triad_ksize(s)


def test_triad_row_and_luts(lib):
    for w, st, so in [(64, 0.35, 0.0), (53, 0.35, 0.5), (130, 0.8, 1.5), (7, 1.0, 4.0), (3840, 0.35, 0.5)]:
        row = tables.triad_row(lib, w, st, so)
        full = orc.make_triad_mask(3, w, st, so)
        assert row.dtype == np.float32 and np.array_equal(row, full[0]) and np.array_equal(row, full[2])
    for g in (2.2, 1.0, 0.5, 3.3):
        a, b = tables.triad_luts(g)
        c, d = orc.triad_luts(g)
        assert np.array_equal(a, c) and np.array_equal(b, d)
    assert not tables.triad_uses_lut(1.0005, False) and tables.triad_uses_lut(1.0005, True)
    assert not tables.triad_uses_lut(0.0, True) and tables.triad_uses_lut(2.2, False)


def test_vignette_warp_scanline_tables():
    for h, w, s in [(48, 64, 0.25), (7, 5, 1.0), (1, 1, 0.5), (1080, 1920, 0.6)]:
        nx2, ny2 = tables.vignette_axes(h, w)
        v = 1.0 - s * np.clip(nx2[None, :] + ny2[:, None], 0.0, 1.0)
        assert np.array_equal(v, orc.make_vignette(h, w, s))
        assert np.array_equal(tables.vignette_full(h, w, s), orc.make_vignette(h, w, s))
    for h, w, s in [(48, 64, 0.15), (33, 47, -0.4), (1, 9, 0.15)]:
        xh, yh, cx, cy = tables.warp_axes(h, w)
        xv, yv = np.meshgrid(xh, yh)
        r2 = xv * xv + yv * yv
        factor = 1.0 + (s * 0.5) * r2
        mx, my = orc.barrel_maps(h, w, s)
        assert np.array_equal((xv * factor * cx + cx).astype(np.float32), mx)
        assert np.array_equal((yv * factor * cy + cy).astype(np.float32), my)
    phases = [0.0, 1.25, 29.0, 1234.5]
    rows = tables.scanline_rows(720, 0.6, 2.0, phases)
    for r, ph in zip(rows, phases):
        assert np.array_equal(r, orc.make_scanline_mask_dynamic(720, 0.6, 2.0, ph))
    assert tables.flicker_factor(0.5, 7.0, 0.3) == float(1.0 + 0.25 * 0.5 * np.sin(2.0 * np.pi * 7.0 * 0.3))


def test_pixelate_maps_match_resize_pair():
    rng = np.random.default_rng(0)
    for h, w, p in [(48, 64, 2), (37, 53, 3), (100, 147, 7), (1080, 1920, 2), (9, 5, 16)]:
        img = rng.random((h, w, 3), dtype=np.float32)
        small = orc.resize(img, (max(1, w // p), max(1, h // p)), "nearest")
        exp = orc.resize(small, (w, h), "nearest")
        xm, ym = tables.pixelate_maps(h, w, p)
        assert np.array_equal(img[ym][:, xm], exp)


def test_resize_axis_matches_oracle_resize():
    rng = np.random.default_rng(1)
    for n_src, n_dst in [(960, 1920), (540, 1080), (26, 53), (53, 26), (7, 64), (1, 9), (480, 1920)]:
        src = rng.random((1, n_src), dtype=np.float32)
        exp = orc.resize(src, (n_dst, 1), "linear")[0]
        ofs, w1 = tables.resize_linear_axis(n_dst, n_src)
        s1 = np.minimum(ofs + 1, n_src - 1)
        got = src[0][ofs] * (np.float32(1.0) - w1) + src[0][s1] * w1
        if n_src == 2 * n_dst:
            continue            # exact 2x decimation takes OpenCV's area path in the oracle, not these taps
        assert np.array_equal(got, exp), (n_src, n_dst)


def test_glitch_offsets_match_oracle():
    for h, w, ph, amp, frac in [(48, 64, 13.0, 9, 0.4), (1080, 1920, 250.0, 40, 0.25), (37, 130, 0.5, 128, 1.0), (20, 20, 3.0, 5, 0.0)]:
        for a, b in ((tables.glitch_offsets_render, orc.glitch_offsets_render), (tables.glitch_offsets_preview, orc.glitch_offsets_preview)):
            y0, o = a(h, w, ph, amp, frac)
            y1, e = b(h, w, ph, amp, frac)
            assert y0 == y1 and ((o is None and e is None) or np.array_equal(o, e))


def test_reference_built_masks_are_recognised():
    """effects._recognise_*: plain arrays with the structure of the reference's builders become descriptors (verified
    element for element, remembered per array object); anything else, and an array rebuilt in place, is looked at again."""
    from pythoncrt_amd import effects
    h, w = 60, 84
    for strength, soft in ((0.35, 0.5), (0.9, 0.0), (0.2, 1.4)):
        tm = orc.make_triad_mask(h, w, strength, soft)
        d = effects._recognise_triad(tm)
        assert isinstance(d, effects.TriadMask) and np.array_equal(np.asarray(d), tm) and effects._recognise_triad(tm) is d
    for s in (0.25, 0.35, 0.123456, 1.0, 0.0):
        vg = orc.make_vignette(h, w, s)
        v = effects._recognise_vignette(vg)
        assert isinstance(v, effects.VignetteMask) and np.array_equal(np.asarray(v), vg), s
    rng = np.random.default_rng(0)
    tm_x = rng.random((h, w, 3)).astype(np.float32)
    assert effects._recognise_triad(tm_x) is tm_x and effects._recognise_triad(tm_x.astype(np.float64)) is not None
    vg_x = orc.make_vignette(h, w, 0.3).copy()
    vg_x[h // 2, w // 2] = 0.5
    assert effects._recognise_vignette(vg_x) is vg_x
    tm = orc.make_triad_mask(h, w, 0.35, 0.5)
    assert isinstance(effects._recognise_triad(tm), effects.TriadMask)
    tm[0, 0, 0] = 0.5                                   # rebuilt in place: no longer row-identical
    assert effects._recognise_triad(tm) is tm
    assert len(effects._RECOGNISED) <= effects._RECOGNISED_MAX


def test_shared_scanline_table_for_integer_phases():
    """pipeline._scan_rows: with integer phases (scanline_speed a multiple of fps) float32(y) + float32(phase) is the
    integer y + phase exactly, so one table over k = y + phase holds every frame's rows bit for bit."""
    from pythoncrt_amd import tables
    h = 270
    for strength, period in ((0.6, 2.0), (0.4, 2.7), (1.0, 1.0)):
        phases = [float(i) for i in range(100, 140)]
        rows = tables.scanline_rows(h, strength, period, phases)
        g = tables.scanline_rows_at(np.arange(100, 139 + h, dtype=np.float32), strength, period)
        assert g.dtype == np.float32 and all(np.array_equal(rows[b], g[b:b + h]) for b in range(len(phases)))


def test_local_states_view():
    """pipeline._LocalStates: what ShardedRender indexes of a chunk's local states (first k frames + the chunk-final one)."""
    import torch
    from pythoncrt_amd.pipeline import _LocalStates
    first = torch.arange(3 * 2 * 2 * 3, dtype=torch.float32).view(3, 2, 2, 3)
    final = torch.full((2, 2, 3), -1.0)
    ls = _LocalStates(first, final, n=10)
    assert ls.shape == (10, 2, 2, 3)
    assert ls[9] is final and ls[-1] is final and torch.equal(ls[1], first[1])
    assert torch.equal(ls[:2], first[:2]) and torch.equal(ls[:3], first)
    with pytest.raises(IndexError):
        ls[:4]
    with pytest.raises(IndexError):
        ls[1:3]


def test_device_state_is_an_array_like():
    """effects.DeviceState — the float state apply_crt_effect returns for numpy frames (ref:699) — behaves as the float32
    array it stands for: np.asarray, indexing, arithmetic, ndarray attributes; writes through it are seen when it is handed
    back (`.tensor`).  (A CPU tensor stands in for the device tensor here.)"""
    import torch
    from pythoncrt_amd.effects import DeviceState, _as_device_tensor
    ref = np.random.default_rng(3).random((5, 7, 3), dtype=np.float32)
    st = DeviceState(torch.from_numpy(ref.copy()))
    assert st.shape == (5, 7, 3) and st.dtype == np.float32 and st.ndim == 3 and st.size == 105 and len(st) == 5
    assert "device-resident" in repr(st)
    assert np.array_equal(np.asarray(st), ref) and np.asarray(st).dtype == np.float32
    assert "materialised" in repr(st)
    assert np.array_equal(np.asarray(st, dtype=np.float64), ref.astype(np.float64))
    assert np.array_equal(st[1:3, 2], ref[1:3, 2]) and st[4, 6, 2] == ref[4, 6, 2]
    assert np.array_equal(st * 2.0, ref * 2.0) and np.array_equal(1.0 - st, 1.0 - ref) and np.array_equal(-st, -ref)
    assert np.array_equal(st + st, ref + ref) and np.array_equal(st > 0.5, ref > 0.5)
    assert np.array_equal(np.abs(st.astype(np.float64) - 0.5), np.abs(ref.astype(np.float64) - 0.5))
    assert float(st.mean()) == float(ref.mean()) and st.tobytes() == ref.tobytes()
    assert np.array_equal(np.clip(st, 0.2, 0.8), np.clip(ref, 0.2, 0.8))
    # handed back as state_prev: the tensor itself, no copy; after a write through the object, the written values
    assert _as_device_tensor(st, torch.device("cpu"), torch.float32, None, "state_prev").data_ptr() == st.tensor.data_ptr()
    st[0, 0, :] = 0.25
    assert np.array_equal(st.tensor.numpy()[0, 0], np.full(3, 0.25, np.float32)) and np.array_equal(st.tensor.numpy()[1:], ref[1:])
    # every array it hands out is read-only: a write the object could not see (the device tensor would go stale) raises instead
    import pytest
    for write in (lambda: np.asarray(st).__setitem__((1, 1, 1), 9.0), lambda: st.numpy().fill(0.0), lambda: st.fill(0.0),
                  lambda: np.clip(st, 0.0, 0.5, out=np.asarray(st)), lambda: st[2].__setitem__(0, 1.0)):
        with pytest.raises(ValueError):
            write()
    assert np.array_equal(st.tensor.numpy()[1:], ref[1:])
    # a copy is an ordinary array: mutate it and pass it back as state_prev (the plain-ndarray path)
    c = np.array(st)
    c[...] = 0.5
    assert c.flags.writeable and np.array_equal(st.tensor.numpy()[1:], ref[1:])


def test_half_quotient_constants_exact():
    """k_phosphor_ct<R, half> normalises a half sample as fma(f, C_HI, f * C_LO) (two instructions behind the conversion) instead of a division.
    With the constants of the kernel source — C_HI = 1/255 rounded down, C_LO = float32(1/255 - C_HI) > 0 — that IS float32(h) / float32(255)
    for every one of the 65 536 half bit patterns: exact rationals for the finite non-zero ones, IEEE special-case rules for zeros (sign kept),
    infinities and NaN."""
    import re
    from fractions import Fraction
    src = open(os.path.join(ROOT, "pythoncrt_amd", "csrc", "crtfx_phosphor_ct.hip.h")).read()
    c_hi = float.fromhex(re.search(r"#define CT_HALF_C_HI (0x[0-9a-fp.+-]+)f", src).group(1))
    c_lo = float.fromhex(re.search(r"#define CT_HALF_C_LO (0x[0-9a-fp.+-]+)f", src).group(1))
    assert np.float32(c_hi) == c_hi and np.float32(c_lo) == c_lo and c_lo > 0 and Fraction(c_hi) < Fraction(1, 255)

    def rn32(fr):
        c = np.float32(float(fr))
        cands = [np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))]
        return np.float32(min(cands, key=lambda x: (abs(Fraction(float(x)) - fr), int(np.float32(x).view(np.uint32)) & 1)))

    hs = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    with np.errstate(all="ignore"):
        ref = hs.astype(np.float32) / np.float32(255.0)
        for h, want in zip(hs, ref):
            f = np.float32(h)
            p = np.float32(f * np.float32(c_lo))                                   # the separately rounded product (-ffp-contract=off)
            if np.isfinite(f) and f != 0:
                got = rn32(Fraction(float(f)) * Fraction(c_hi) + Fraction(float(p)))
                assert got.view(np.uint32) == want.view(np.uint32), float(h)
            elif np.isnan(f):
                assert np.isnan(want)
            elif f == 0:
                assert p == 0 and np.signbit(p) == np.signbit(f) and np.signbit(want) == np.signbit(f)      # (+-0) * C_HI + (+-0): the sign survives only because C_LO > 0
            else:
                assert np.isinf(p) and np.signbit(p) == np.signbit(f) and want == f                           # inf * C_HI + inf (same sign): inf, not NaN


def test_option_ids_match_the_header():
    """effects._OPTION_IDS (the names bench.py --opt and the tests use) against the CRTFX_OPT_* enumerators of include/crtfx.h: one table cannot drift
    from the other (round 6 added NO_FUSED_HALF = 16)."""
    import os
    import re
    from pythoncrt_amd import effects
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "crtfx.h")).read()
    enum = {m.group(1): int(m.group(2)) for m in re.finditer(r"CRTFX_OPT_([A-Z_]+)\s*=\s*(\d+)", text)}
    assert enum and enum == effects._OPTION_IDS, (sorted(set(enum.items()) ^ set(effects._OPTION_IDS.items())))
