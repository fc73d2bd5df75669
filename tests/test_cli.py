"""pythoncrt_amd.cli against the reference's own CLI (tests/golden/reference_cli.json, made by
gen_golden.py from parse_args ref:1153-1207 and main ref:1210-1267): same flag names and
defaults, same clamps on the way to the render settings."""
import json
import os

import numpy as np

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
    assert set(ours) - set(ref) == {"batch", "noise_seed", "staging_report", "io"}      # the additions documented in cli.py


def test_clamps_match_reference_main():
    for name, argv in ARGVS.items():
        rs = cli.settings_from_args(cli.build_parser().parse_args(argv))
        ref = FX["process_video_kwargs"][name]
        for k, v in ref.items():
            if k in RENAME:
                continue
            assert getattr(rs, k) == v, (name, k, getattr(rs, k), v)


def test_positional_file_io_helpers(tmp_path, monkeypatch):
    """cli._pread_full / _pwrite_full: sliced, threaded positional I/O equals one plain read / write — offsets, a short file, sizes that
    are not a multiple of the slice."""
    import numpy as np
    monkeypatch.setattr(cli, "_IO_SLICE", 1000)                      # many slices on a small file
    data = np.random.default_rng(5).integers(0, 256, 12345, dtype=np.uint8).tobytes()
    src = tmp_path / "src.bin"
    src.write_bytes(data)
    fd = os.open(src, os.O_RDONLY)
    try:
        buf = bytearray(5000)
        assert cli._pread_full(fd, memoryview(buf), 2345) == 5000 and bytes(buf) == data[2345:7345]
        buf = bytearray(4000)                                        # runs past the end of the file: the contiguous prefix only
        got = cli._pread_full(fd, memoryview(buf), 10000)
        assert got == 2345 and bytes(buf[:got]) == data[10000:]
        buf = bytearray(300)                                         # a single slice
        assert cli._pread_full(fd, memoryview(buf), 0) == 300 and bytes(buf) == data[:300]
    finally:
        os.close(fd)
    dst = tmp_path / "dst.bin"
    fd = os.open(dst, os.O_WRONLY | os.O_CREAT)
    try:
        cli._pwrite_full(fd, memoryview(data)[:7000], 0)
        cli._pwrite_full(fd, memoryview(data)[7000:], 7000)
        cli._pwrite_full(fd, memoryview(b""), 0)
    finally:
        os.close(fd)
    assert dst.read_bytes() == data
    with open(src, "rb") as f:
        assert cli._seekable(f)
    r, w = os.pipe()
    try:
        with os.fdopen(r, "rb") as pr:
            assert not cli._seekable(pr)
    finally:
        os.close(w)


def test_mapped_input_reads_like_the_file(tmp_path):
    """cli._MappedInput — the regular-file input path (mmap + memcpy on the I/O threads instead of read(2)): whole batches, the short
    last batch, offsets past the end, slices larger and smaller than the I/O slice, an empty file."""
    import os
    data = np.random.default_rng(0).integers(0, 256, 40_000_003, dtype=np.uint8)
    path = tmp_path / "in.rgb"
    path.write_bytes(data.tobytes())
    fd = os.open(path, os.O_RDONLY)
    m = cli._MappedInput(fd)
    dst = np.empty(33_000_000, dtype=np.uint8)           # > 16 MiB: split over the pool
    assert m.read_into(dst, 0) == dst.size and np.array_equal(dst, data[:dst.size])
    assert m.read_into(dst, 33_000_000) == 7_000_003 and np.array_equal(dst[:7_000_003], data[33_000_000:])
    assert m.read_into(dst, data.size) == 0 and m.read_into(dst, data.size + 5) == 0
    small = np.empty(1000, dtype=np.uint8)
    assert m.read_into(small, 12345) == 1000 and np.array_equal(small, data[12345:13345])
    m.close()
    os.close(fd)
    empty = tmp_path / "empty.rgb"
    empty.write_bytes(b"")
    fd = os.open(empty, os.O_RDONLY)
    m = cli._MappedInput(fd)
    assert m.map is None and m.read_into(small, 0) == 0
    m.close()
    os.close(fd)


def test_cli_pipes_get_the_large_buffer():
    """cli._grow_pipe: F_SETPIPE_SZ on both ends of a pipe (1 MiB, or the most the kernel grants); 0 for a regular file."""
    from pythoncrt_amd import cli
    r, wfd = os.pipe()
    try:
        with os.fdopen(r, "rb", buffering=0) as fr, os.fdopen(wfd, "wb", buffering=0) as fw:
            got = cli._grow_pipe(fw)
            assert got >= 65536 and cli._grow_pipe(fr) == got          # one buffer, seen from both ends
        with open(__file__, "rb") as f:
            assert cli._grow_pipe(f) == 0
    finally:
        pass
