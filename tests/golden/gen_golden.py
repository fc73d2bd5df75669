#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own function bodies.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/gen_golden.py

How the reference is run: `crt_filter.py` is read as TEXT, parsed with `ast`, and only the
`FunctionDef` nodes named in WANT are compiled into a private namespace that holds numpy, the
three perf globals and a *recording* `cv2` object.  The module top level is never executed (its
import-time `ensure_deps()` at crt_filter.py:47 would shell out to `pip install`, and cv2 /
moviepy / PySide6 are absent here), and no reference source text is stored in the fixtures: a
fixture is inputs + the outputs those function bodies produced.

The recording `cv2` does NOT emulate OpenCV.  Its methods store the arguments they were
called with (so the fixtures can pin the kernel sizes, the float32 remap maps and the
pre-quantise float image the reference hands to OpenCV) and return an input unchanged so the
surrounding numpy code keeps running.  Outputs downstream of such a call are never stored as
expected values, except where the call is provably the last statement before the captured value.
"""
from __future__ import annotations

import ast
import os
import threading
import time
from collections import defaultdict
from typing import Optional, Tuple

import numpy as np

REF = "/root/reference/crt_filter.py"
OUT = os.path.dirname(os.path.abspath(__file__))

WANT = {
    "perf_add", "shift_channel", "make_scanline_mask_dynamic", "make_triad_mask", "_apply_triad_mask",
    "make_vignette", "apply_color_adjustments", "make_scanline_mask_2d", "apply_barrel_warp",
    "apply_static_effects", "apply_crt_effect",
}


class RecordingCv2:
    """Capture-only stand-in: records call arguments, computes nothing."""
    INTER_NEAREST, INTER_LINEAR, BORDER_REPLICATE, BORDER_CONSTANT = 0, 1, 1, 0

    def __init__(self):
        self.calls = []

    def GaussianBlur(self, src, ksize, sigmaX=0, sigmaY=0, borderType=None):
        self.calls.append(("GaussianBlur", tuple(ksize), float(sigmaX), float(sigmaY), borderType, np.array(src, copy=True)))
        return src

    def remap(self, img, map_x, map_y, interpolation=None, borderMode=None, borderValue=None):
        self.calls.append(("remap", np.array(map_x, copy=True), np.array(map_y, copy=True), interpolation, borderMode, borderValue))
        return img

    def convertScaleAbs(self, img, alpha=1.0, beta=0.0):
        self.calls.append(("convertScaleAbs", np.array(img, copy=True), float(alpha), float(beta)))
        return np.zeros(img.shape, np.uint8)


def load_reference():
    tree = ast.parse(open(REF).read())
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    missing = WANT - {n.name for n in nodes}
    assert not missing, missing
    cv2 = RecordingCv2()
    ns = {
        "np": np, "time": time, "cv2": cv2, "Image": None, "Optional": Optional, "Tuple": Tuple,
        "_perf_lock": threading.Lock(), "_perf_totals": defaultdict(float), "_perf_counts": defaultdict(int),
    }
    exec(compile(ast.Module(body=nodes, type_ignores=[]), "<reference functions>", "exec"), ns)
    return ns, cv2


CLI_ARGVS = {
    "defaults": ["--input", "x.mp4"],
    "extremes_hi": ["--input", "x.mp4", "--scanline-strength", "5", "--triad-strength", "3", "--triad-gamma", "0.01", "--triad-softness", "-1",
                    "--aberration-px", "40", "--bloom-sigma", "-2", "--bloom-strength", "-1", "--noise-strength", "-3", "--vignette-strength", "9",
                    "--persistence", "0.99", "--scanline-period", "0.2", "--pixel-size", "0", "--gamma", "0", "--saturation", "-1",
                    "--temperature", "4", "--flicker-strength", "7", "--flicker-hz", "-1", "--grain-size", "0", "--scanline-thickness", "0.01",
                    "--warp-strength", "3", "--glitch-amp", "-5", "--glitch-height", "2", "--bloom-threshold", "1.5", "--crf", "99"],
    "extremes_lo": ["--input", "x.mp4", "--scanline-strength", "-1", "--aberration-px", "-40", "--persistence", "-0.5", "--temperature", "-4",
                    "--warp-strength", "-3", "--vignette-strength", "-1", "--bloom-threshold", "-1", "--crf", "1", "--no-fast-bloom",
                    "--triad-preserve-luma", "--text-after", "--fps", "25", "--width", "640", "--height", "360"],
}


def dump_cli_fixture():
    """parse_args (ref:1153-1207) and the clamps of main (ref:1220-1267): the flag table and what
    main() hands to process_video for a few argv sets.  process_video / Path are recording stubs."""
    import argparse
    import json
    import sys
    tree = ast.parse(open(REF).read())
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("parse_args", "main")]
    captured = {}

    class FakePath:
        def __init__(self, s): self.s = str(s); self.stem = "x"
        def exists(self): return True
        def with_name(self, n): return FakePath(n)
        def __str__(self): return self.s

    def process_video(**kw):
        captured.clear()
        captured.update({k: (str(v) if isinstance(v, FakePath) else v) for k, v in kw.items()})
        return False

    ns = {"argparse": argparse, "Path": FakePath, "time": time, "process_video": process_video,
          "launch_gui": lambda: None, "print": lambda *a, **k: None}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), "<reference cli>", "exec"), ns)
    fixture = {"process_video_kwargs": {}}
    old = sys.argv
    try:
        for name, argv in CLI_ARGVS.items():
            sys.argv = ["crt_filter.py"] + argv
            ns["main"]()
            fixture["process_video_kwargs"][name] = dict(captured)
        sys.argv = ["crt_filter.py"]
        fixture["namespace_defaults"] = vars(ns["parse_args"]())
    finally:
        sys.argv = old
    json.dump(fixture, open(os.path.join(OUT, "reference_cli.json"), "w"), indent=1, sort_keys=True)


TEXT_CASES = {       # name -> (w, h, text, font_family, size, color_hex, pos)
    "default_font": (320, 96, "Hello CRT 0123", "", 36, "#FFCC00", (32, 32)),
    "ttf_path": (256, 80, "AMD gfx950", "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf", 28, "33aaff", (-6, 10)),
    "unknown_family": (200, 64, "phosphor", "No Such Face", 20, "#zzzzzz", (4, 40)),
    "empty": (40, 20, "", "", 36, "#FFFFFF", (32, 32)),
}
HEX_CASES = ["#FFCC00", "33aaff", " #010203 ", "#fff", "", "#zzzzzz", "12345", "#1234567"]


def dump_text_fixture():
    """The PIL text rasteriser (ref:350-414) and the size-mismatch resize of the overlay inside the
    chain (ref:758-767): outputs of the reference's own function bodies on this image's Pillow."""
    from PIL import Image, ImageDraw, ImageFont
    tree = ast.parse(open(REF).read())
    want = {"_parse_hex_color", "_make_text_overlay_rgba", "perf_add", "shift_channel", "apply_color_adjustments",
            "make_scanline_mask_dynamic", "make_scanline_mask_2d", "_apply_triad_mask", "apply_barrel_warp", "apply_static_effects"}
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert want == {n.name for n in nodes}
    ns = {"np": np, "os": os, "time": time, "cv2": RecordingCv2(), "Image": Image, "ImageDraw": ImageDraw, "ImageFont": ImageFont,
          "Optional": Optional, "Tuple": Tuple, "_perf_lock": threading.Lock(), "_perf_totals": defaultdict(float),
          "_perf_counts": defaultdict(int)}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), "<reference functions>", "exec"), ns)
    out = {}
    for name, (w, h, text, fam, size, col, pos) in TEXT_CASES.items():
        if fam.startswith("/") and not os.path.isfile(fam):
            continue
        out[f"text/{name}"] = ns["_make_text_overlay_rgba"](w, h, text, fam, size, col, pos)
    out["hex"] = np.array([ns["_parse_hex_color"](c) for c in HEX_CASES], dtype=np.int64)
    # overlay smaller than the frame, blended before and after an otherwise empty chain
    ov = ns["_make_text_overlay_rgba"](64, 24, "ab", "", 36, "#80FF40", (2, 2))
    rng = np.random.default_rng(77)
    frame = rng.integers(0, 256, (36, 80, 3), dtype=np.uint8)
    out["fit/overlay"], out["fit/frame"] = ov, frame
    for after in (False, True):
        out[f"fit/static_after{int(after)}"] = ns["apply_static_effects"](
            frame, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0, 0.0,
            text_overlay_rgba=ov, text_overlay_after=after)
    np.savez_compressed(os.path.join(OUT, "reference_text_overlay.npz"), **out)
    print(f"wrote {len(out)} text arrays")


def frames(h, w):
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = np.stack([(xx * 255) // max(1, w - 1), (yy * 255) // max(1, h - 1), ((xx + yy) * 255) // max(1, h + w - 2)], axis=2).astype(np.uint8)
    imp = np.zeros((h, w, 3), np.uint8)
    imp[h // 2, w // 2] = 255
    return {"noise": noise, "grad": grad, "impulse": imp}


def main():
    ns, cv2 = load_reference()
    out = {}

    # ---- a2 shift_channel -------------------------------------------------------------
    plane = np.random.default_rng(1).random((12, 17), dtype=np.float32)
    for dx in (-8, -1, 1, 3, 8):
        out[f"shift/dx{dx}"] = ns["shift_channel"](plane, dx, 0)
    out["shift/in"] = plane

    # ---- a8 scanline masks ------------------------------------------------------------
    scan1d = [(48, 0.6, 2.0, 0.0), (48, 0.6, 2.0, 1.25), (96, 1.0, 3.7, 12.5), (33, 0.25, 1.0, 0.5), (720, 0.6, 2.0, 29.0)]
    for i, (h, s, p, ph) in enumerate(scan1d):
        out[f"scan1d/{i}"] = ns["make_scanline_mask_dynamic"](h, s, p, ph)
    out["scan1d/args"] = np.array(scan1d, np.float64)
    scan2d = [(24, 40, 0.6, 2.0, 1.25, 5.0, 1.0), (24, 40, 0.8, 3.0, 0.0, -12.5, 2.5), (16, 16, 0.5, 2.0, 7.0, 0.0, 0.3)]
    for i, a in enumerate(scan2d):
        out[f"scan2d/{i}"] = ns["make_scanline_mask_2d"](int(a[0]), int(a[1]), *a[2:])
    out["scan2d/args"] = np.array(scan2d, np.float64)

    # ---- a6 triad mask (softness 0 is numpy-only; softness>0 pins only the blur call args) ---
    out["triad_mask/s0"] = ns["make_triad_mask"](5, 20, 0.35, 0.0)
    soft = [0.2, 0.5, 0.83, 0.84, 1.0, 1.5, 2.5, 4.0]
    ks = []
    for s in soft:
        cv2.calls.clear()
        ns["make_triad_mask"](3, 12, 0.35, s)
        ks.append(cv2.calls[0][1])
        assert cv2.calls[0][2] == s and cv2.calls[0][3] == 0.0 and cv2.calls[0][4] == cv2.BORDER_REPLICATE
    out["triad_mask/soft"] = np.array(soft)
    out["triad_mask/soft_ksize"] = np.array(ks, np.int64)

    # ---- a9 vignette --------------------------------------------------------------------
    for i, (h, w, s) in enumerate([(48, 64, 0.25), (7, 5, 1.0), (1, 1, 0.5), (96, 128, 0.6)]):
        out[f"vignette/{i}"] = ns["make_vignette"](h, w, s)
    out["vignette/args"] = np.array([(48, 64, 0.25), (7, 5, 1.0), (1, 1, 0.5), (96, 128, 0.6)], np.float64)

    # ---- a4 colour grade ----------------------------------------------------------------
    fr = frames(48, 64)
    img0 = fr["noise"].astype(np.float32) / 255.0
    grades = [
        (0.0, 1.0, 1.0, 1.0, 0.0), (0.1, 1.2, 1.0, 1.0, 0.0), (0.0, 1.0, 2.2, 1.0, 0.0), (0.0, 1.0, 1.0, 1.6, 0.0),
        (0.0, 1.0, 1.0, 0.0, 0.0), (0.0, 1.0, 1.0, 1.0, 0.7), (0.0, 1.0, 1.0, 1.0, -1.0), (-0.05, 0.8, 0.6, 1.3, -0.4),
    ]
    for i, (b, c, g, s, t) in enumerate(grades):
        out[f"grade/{i}"] = ns["apply_color_adjustments"](img0.copy(), b, c, g, s, t)
    out["grade/args"] = np.array(grades, np.float64)
    out["grade/in_u8"] = fr["noise"]

    # ---- a7 triad apply -----------------------------------------------------------------
    mask = ns["make_triad_mask"](48, 64, 0.35, 0.0)
    tri = [(2.2, False), (2.2, True), (1.0, False), (1.0, True), (0.5, True), (1.0005, False), (0.0, True), (3.3, False)]
    for i, (g, p) in enumerate(tri):
        out[f"triad_apply/{i}"] = ns["_apply_triad_mask"](img0.copy(), mask, g, p)
    out["triad_apply/args"] = np.array([(g, float(p)) for g, p in tri], np.float64)
    gi = frames(48, 64)["grad"].astype(np.float32) / 255.0
    out["triad_apply/grad_2.2_T"] = ns["_apply_triad_mask"](gi, mask, 2.2, True)

    # ---- a12 barrel maps (numpy part of apply_barrel_warp; remap args recorded) -----------
    for i, (h, w, s) in enumerate([(48, 64, 0.15), (33, 47, -0.4), (96, 128, 1.0), (1, 9, 0.15), (1080, 16, 0.15)]):
        cv2.calls.clear()
        ns["apply_barrel_warp"](np.zeros((h, w, 3), np.float32), s)
        c = cv2.calls[0]
        assert c[0] == "remap" and c[3] == cv2.INTER_LINEAR and c[4] == cv2.BORDER_CONSTANT and c[5] == 0
        out[f"warpmap/{i}/x"], out[f"warpmap/{i}/y"] = c[1], c[2]
    out["warpmap/args"] = np.array([(48, 64, 0.15), (33, 47, -0.4), (96, 128, 1.0), (1, 9, 0.15), (1080, 16, 0.15)], np.float64)

    # ---- a5 bloom ksize table (GaussianBlur call args recorded) ---------------------------
    sig = [0.1, 0.17, 0.5, 0.83, 0.84, 1.2, 1.5, 2.5, 3.0, 3.5, 10.0]
    bk = []
    for s in sig:
        cv2.calls.clear()
        ns["apply_static_effects"](fr["noise"][:8, :8], 0.0, None, 2.2, False, 0, s, 0.25, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0)
        c = cv2.calls[0]
        assert c[0] == "GaussianBlur" and c[2] == s and c[3] == s and c[4] == cv2.BORDER_REPLICATE
        bk.append(c[1])
    out["bloom/sigma"] = np.array(sig)
    out["bloom/ksize"] = np.array(bk, np.int64)
    # the thresholded source image handed to the blur (numpy-only up to that call)
    cv2.calls.clear()
    ns["apply_static_effects"](fr["noise"], 0.0, None, 2.2, False, 1, 3.0, 0.25, 0.4, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0,
                               brightness=0.05, contrast=1.1)
    out["bloom/src_thr0.4"] = cv2.calls[0][5]

    # ---- a17 chain: apply_static_effects on cv2-free parameter sets ------------------------
    chains = {
        # BASELINE config 1: scanlines only (phase = i/30*30 for i = 0 and 7)
        "scan_only_p0": dict(scanline_strength=0.6, triad=None, vig=None, aberration_px=0, scanline_phase_px=0.0),
        "scan_only_p7": dict(scanline_strength=0.6, triad=None, vig=None, aberration_px=0, scanline_phase_px=7.0),
        # everything numpy-only switched on
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
    for size in ((48, 64), (96, 128)):
        f = frames(*size)
        for fname in ("noise", "grad"):
            for cname, c in chains.items():
                h, w = size
                if size != (48, 64) and not (fname == "noise" and cname in ("numpy_full", "glitch_render")):
                    continue  # keep the fixture file small: the larger size only re-checks two chains
                tm = ns["make_triad_mask"](h, w, *c["triad"]) if c.get("triad") else None
                vg = ns["make_vignette"](h, w, c["vig"]) if c.get("vig") else None
                res = ns["apply_static_effects"](
                    f[fname], c["scanline_strength"], tm, c.get("triad_gamma", 2.2), c.get("triad_preserve_luma", False),
                    c["aberration_px"], 0.0, 0.0, 0.0, 0.0, vg, 2.0, c["scanline_phase_px"], False, 1,
                    c.get("glitch_amp_px", 0), c.get("glitch_height_frac", 0.0), time_sec=c.get("time_sec", 0.0),
                    brightness=c.get("brightness", 0.0), contrast=c.get("contrast", 1.0), gamma=c.get("gamma", 1.0),
                    saturation=c.get("saturation", 1.0), temperature=c.get("temperature", 0.0),
                    flicker_strength=c.get("flicker_strength", 0.0), flicker_hz=c.get("flicker_hz", 0.0),
                    scanline_angle=c.get("scanline_angle", 0.0), scanline_thickness=c.get("scanline_thickness", 1.0))
                out[f"chain/{h}x{w}/{fname}/{cname}"] = res
    # preview-path glitch (ref:664-686): float image captured at the convertScaleAbs call
    f = frames(48, 64)
    cv2.calls.clear()
    ns["apply_crt_effect"](f["noise"], 0.6, None, 2.2, False, 1, 0.0, 0.0, 0.0, 0.0, None, 0.0, None, 2.0, 250.0, False, 1,
                           glitch_amp_px=11, glitch_height_frac=0.5)
    c = cv2.calls[-1]
    assert c[0] == "convertScaleAbs" and c[2] == 255.0 and c[3] == 0.0
    out["chain/48x64/noise/glitch_preview_float"] = c[1]

    dump_cli_fixture()
    dump_text_fixture()
    np.savez_compressed(os.path.join(OUT, "reference_numpy_stages.npz"), **out)
    total = sum(v.nbytes for v in out.values())
    print(f"wrote {len(out)} arrays, {total/1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
