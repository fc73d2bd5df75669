"""Host-side text overlay: the RGBA plane the chain alpha-blends before or after the
effects.  Rasterisation stays on the CPU with Pillow, as in the reference's PIL path
(crt_filter.py `_parse_hex_color` ref:350-363, `_make_text_overlay_rgba` ref:366-414 — the Qt painter
variant ref:417-466 falls back to this one when PySide6 is absent, and the GUI itself is out of scope).
Pinned against the reference's function on this image's Pillow by tests/test_text_overlay.py.
"""
from __future__ import annotations

import os
from typing import Sequence, Tuple

import numpy as np


def parse_hex_color(s) -> Tuple[int, int, int]:
    """'#RRGGBB' or 'RRGGBB' -> (r, g, b); anything else is white (ref:350-363)."""
    try:
        body = s.strip()
        body = body[1:] if body.startswith("#") else body
        if len(body) == 6:
            return tuple(int(body[i:i + 2], 16) for i in (0, 2, 4))
    except Exception:
        pass
    return (255, 255, 255)


# family name -> file name under %WINDIR%\Fonts (ref:386-393)
_WINDOWS_FACES = {"arial": "arial.ttf", "segoe ui": "segoeui.ttf", "consolas": "consola.ttf", "tahoma": "tahoma.ttf",
                  "times new roman": "times.ttf", "courier new": "cour.ttf"}


def _resolve_font(font_family: str, size: int):
    """Lookup order of ref:372-410: a font FILE path, the Windows face table, '<family>.ttf' in the Windows
    font directory, Pillow's own search for arial.ttf, Pillow's built-in bitmap font."""
    from PIL import ImageFont

    def try_file(path):
        try:
            return ImageFont.truetype(path, size) if os.path.isfile(path) else None
        except Exception:
            return None

    font = try_file(font_family) if font_family else None
    if font is None:
        fam = (font_family or "").lower()
        fonts_dir = os.path.join(os.environ.get("WINDIR", "C:\\Windows"), "Fonts")
        names = ([_WINDOWS_FACES[fam]] if fam in _WINDOWS_FACES else []) + ([fam + ".ttf"] if fam else [])
        for name in names:
            font = try_file(os.path.join(fonts_dir, name))
            if font is not None:
                break
    if font is None:
        try:
            font = ImageFont.truetype("arial.ttf", size)
        except Exception:
            font = ImageFont.load_default()
    return font


def make_text_overlay_rgba(w: int, h: int, text: str, font_family: str = "", size: int = 36, color_hex: str = "#FFFFFF",
                           pos: Sequence[int] = (32, 32)) -> np.ndarray:
    """H x W x 4 uint8: `text` drawn opaque in `color_hex` with its top-left at `pos` on a transparent
    plane; all zeros for an empty string (ref:366-414)."""
    if not text:
        return np.zeros((h, w, 4), dtype=np.uint8)
    from PIL import Image, ImageDraw
    plane = Image.new("RGBA", (w, h), (0, 0, 0, 0))
    ImageDraw.Draw(plane).text((int(pos[0]), int(pos[1])), text, font=_resolve_font(font_family, size),
                               fill=parse_hex_color(color_hex) + (255,))
    return np.asarray(plane, dtype=np.uint8)


def fit_overlay(ov: np.ndarray, h: int, w: int) -> np.ndarray:
    """An overlay whose size differs from the frame's is resampled by Pillow's bilinear filter (ref:593-594,
    :658-659) — on the host, with the same library call, so the result is the reference's by construction."""
    if ov.shape[0] == h and ov.shape[1] == w:
        return ov
    from PIL import Image
    return np.asarray(Image.fromarray(ov, mode="RGBA").resize((w, h), Image.BILINEAR))
