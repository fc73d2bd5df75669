"""SURVEY 8f row 1, host half: the Pillow text rasteriser and the overlay size fit, pinned against the
outputs of the reference's own function bodies (tests/golden/gen_golden.py: dump_text_fixture)."""
import os

import numpy as np
import pytest

from pythoncrt_amd import text
from oracle import crt_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
import importlib.util
_spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "golden", "gen_golden.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


@pytest.fixture(scope="module")
def tg():
    return np.load(os.path.join(HERE, "golden", "reference_text_overlay.npz"))


def test_parse_hex_color(tg):
    got = np.array([text.parse_hex_color(c) for c in gen.HEX_CASES], dtype=np.int64)
    assert np.array_equal(got, tg["hex"])
    assert text.parse_hex_color(None) == (255, 255, 255)


@pytest.mark.parametrize("name", list(gen.TEXT_CASES))
def test_rasteriser_matches_reference(tg, name):
    if f"text/{name}" not in tg.files:
        pytest.skip("fixture was generated without this font file")
    w, h, txt, fam, size, col, pos = gen.TEXT_CASES[name]
    if fam.startswith("/") and not os.path.isfile(fam):
        pytest.skip("font file not on this machine")
    got = text.make_text_overlay_rgba(w, h, txt, fam, size, col, pos)
    exp = tg[f"text/{name}"]
    assert got.dtype == np.uint8 and got.shape == exp.shape == (h, w, 4)
    assert np.array_equal(got, exp)
    if txt:
        assert exp[..., 3].max() > 0             # something was drawn


def test_fit_overlay_and_oracle_chain(tg):
    ov, frame = tg["fit/overlay"], tg["fit/frame"]
    h, w = frame.shape[:2]
    fitted = text.fit_overlay(ov, h, w)
    assert fitted.shape == (h, w, 4) and fitted.dtype == np.uint8
    assert text.fit_overlay(fitted, h, w) is fitted
    for after in (False, True):
        exp = tg[f"fit/static_after{int(after)}"]
        got = orc.apply_static_effects(frame, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0, 0.0,
                                       text_overlay_rgba=ov, text_overlay_after=after)
        assert got.dtype == exp.dtype and np.array_equal(got, exp)
        # the same blend with the overlay fitted by the product's host helper
        got2 = orc.apply_static_effects(frame, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0, 0.0,
                                        text_overlay_rgba=fitted, text_overlay_after=after)
        assert np.array_equal(got2, exp)
