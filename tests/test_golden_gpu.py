"""GPU: the HIP path against the reference's OWN outputs, with no oracle in the loop.  tests/golden/
reference_numpy_stages.npz holds what the reference's function bodies returned for whole chains made of the numpy-only
stages (aberration, colour grade, triad + LUTs, 1-D / 2-D scanlines, vignette, flicker, glitch); the same calls go
through pythoncrt_amd here.  Tolerance as in tests/test_oracle_golden.py: 2.4e-7 absolute (sin / pow tables, the
reference's float64 tail against the float32 result), and the values the reference hands to convertScaleAbs in the
preview path."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("oracle_golden", os.path.join(HERE, "test_oracle_golden.py"))
og = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(og)


@pytest.fixture(scope="module")
def pc():
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    import pythoncrt_amd
    return pythoncrt_amd


def run_chain_gpu(pc, frame, c):
    h, w = frame.shape[:2]
    tm = pc.make_triad_mask(h, w, *c["triad"]) if c.get("triad") else None
    vg = pc.make_vignette(h, w, c["vig"]) if c.get("vig") else None
    return pc.apply_static_effects(
        frame, c["scanline_strength"], tm, c.get("triad_gamma", 2.2), c.get("triad_preserve_luma", False),
        c["aberration_px"], 0.0, 0.0, 0.0, 0.0, vg, 2.0, c["scanline_phase_px"], False, 1,
        c.get("glitch_amp_px", 0), c.get("glitch_height_frac", 0.0), time_sec=c.get("time_sec", 0.0),
        brightness=c.get("brightness", 0.0), contrast=c.get("contrast", 1.0), gamma=c.get("gamma", 1.0),
        saturation=c.get("saturation", 1.0), temperature=c.get("temperature", 0.0),
        flicker_strength=c.get("flicker_strength", 0.0), flicker_hz=c.get("flicker_hz", 0.0),
        scanline_angle=c.get("scanline_angle", 0.0), scanline_thickness=c.get("scanline_thickness", 1.0))


@pytest.mark.parametrize("cname", sorted(og.CHAINS))
def test_chain_against_reference_outputs(pc, golden, cname):
    keys = [k for k in golden.files if k.startswith("chain/") and k.endswith("/" + cname)]
    assert keys
    for k in keys:
        _, size, fname, _ = k.split("/")
        h, w = map(int, size.split("x"))
        got = run_chain_gpu(pc, og.frames(h, w)[fname], og.CHAINS[cname])
        exp = golden[k]
        assert got.dtype == np.float32 and got.shape == exp.shape
        assert np.abs(got.astype(np.float64) - exp.astype(np.float64)).max() <= 2.4e-7, k


def test_preview_glitch_against_reference_output(pc, golden):
    f = og.frames(48, 64)["noise"]
    out, state = pc.apply_crt_effect(f, 0.6, None, 2.2, False, 1, 0.0, 0.0, 0.0, 0.0, None, 0.0, None, 2.0, 250.0, False, 1,
                                     glitch_amp_px=11, glitch_height_frac=0.5)
    exp = golden["chain/48x64/noise/glitch_preview_float"]          # the float image the reference passes to convertScaleAbs
    assert np.abs(state.astype(np.float64) - exp.astype(np.float64)).max() <= 2.4e-7
    q = np.rint(np.abs(exp.astype(np.float32) * np.float32(255.0))).clip(0, 255).astype(np.uint8)      # cv2.convertScaleAbs
    d = np.abs(out.astype(np.int16) - q.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3


def test_normalise_and_aberration_against_numpy(pc):
    """a1 + a2 alone: true division by 255 and the wrap-around shifts, against numpy's own roll (ref:207-210, 571-577)."""
    f = og.frames(48, 64)
    off = lambda fr, **kw: pc.apply_static_effects(fr, 0.0, None, 2.2, False, kw.pop("ab", 0), 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0, **kw)
    img = f["noise"].astype(np.float32) / 255.0
    for d in (-8, -1, 1, 3, 8):
        got = off(f["noise"], ab=d)
        exp = np.stack([np.roll(img[:, :, 0], d, 1), img[:, :, 1], np.roll(img[:, :, 2], -d, 1)], axis=2)   # ref:571-577
        assert np.array_equal(got, exp)
