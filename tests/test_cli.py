"""pythoncrt_amd.cli against the reference's own CLI (tests/golden/reference_cli.json, made by
gen_golden.py from parse_args ref:1153-1207 and main ref:1210-1267): same flag names and
defaults, same clamps on the way to the render settings."""
import json
import os

from pythoncrt_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_cli.json")))
ARGVS = {
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
# process_video keyword -> RenderSettings field
RENAME = {"target_bitrate_kbps": None, "input_path": None, "output_path": None, "width": None, "height": None, "fps": None, "crf": None,
          "gpu": None, "nvenc_preset": None, "encoder_preference": None, "decoder_preference": None, "text": None, "text_font": None,
          "text_size": None, "text_color": None, "text_pos": None, "text_after": None}


def test_flag_names_and_defaults():
    ours = vars(cli.build_parser().parse_args([]))
    ref = FX["namespace_defaults"]
    for k, v in ref.items():
        assert k in ours, k
        assert ours[k] == v, (k, ours[k], v)
    assert set(ours) - set(ref) == {"batch", "noise_seed"}      # the two additions documented in cli.py


def test_clamps_match_reference_main():
    for name, argv in ARGVS.items():
        rs = cli.settings_from_args(cli.build_parser().parse_args(argv))
        ref = FX["process_video_kwargs"][name]
        for k, v in ref.items():
            if k in RENAME:
                continue
            assert getattr(rs, k) == v, (name, k, getattr(rs, k), v)
