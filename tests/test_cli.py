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


def test_registered_map_windows(tmp_path, monkeypatch):
    """cli._RegisteredMap (the --io mapped path) without a GPU: hipHostRegister / Unregister / MemcpyAsync replaced by recorders.  Windows sit on a
    fixed grid, a window two batches share is registered once and unregistered when the second one lets go, a refused registration leaves nothing
    registered for that call, and a copy is issued window by window (the runtime rejects one that runs across two registrations)."""
    calls = []
    refuse = set()

    class FakeDma:
        H2D, D2H = 1, 2

        @staticmethod
        def register(ptr, n):
            if ptr in refuse:
                return False
            calls.append(("reg", ptr, n))
            return True

        @staticmethod
        def unregister(ptr):
            calls.append(("unreg", ptr))

        @staticmethod
        def copy(dst, src, n, kind, stream):
            calls.append(("copy", dst, src, n, kind))
    monkeypatch.setattr(cli, "_HostDma", FakeDma)
    monkeypatch.setattr(cli._RegisteredMap, "WIN", 64 * 1024)
    win = 64 * 1024
    size = 5 * win + 1234
    path = tmp_path / "f.bin"
    path.write_bytes(os.urandom(size))
    fd = os.open(path, os.O_RDONLY)
    try:
        m = cli._RegisteredMap(fd, size, writable=False)
        base = m.base
        # batch A: bytes [10 000, 150 000) = windows 0, 1, 2
        assert m.acquire(10_000, 140_000)
        assert [c for c in calls if c[0] == "reg"] == [("reg", base, win), ("reg", base + win, win), ("reg", base + 2 * win, win)]
        # batch B: [150 000, 300 000) = windows 2, 3, 4: window 2 is shared and not registered again
        calls.clear()
        assert m.acquire(150_000, 150_000)
        assert [c[1] for c in calls if c[0] == "reg"] == [base + 3 * win, base + 4 * win]
        # the copy of batch B runs window by window, contiguous on both sides
        calls.clear()
        m.copy(FakeDma.H2D, 1 << 40, 150_000, 150_000, 0)
        cp = [c for c in calls if c[0] == "copy"]
        assert [c[3] for c in cp] == [3 * win - 150_000, win, 300_000 - 4 * win] and sum(c[3] for c in cp) == 150_000
        assert cp[0][1] == 1 << 40 and cp[0][2] == base + 150_000 and cp[1][1] == (1 << 40) + cp[0][3] and cp[1][2] == base + 3 * win
        calls.clear()
        m.copy(FakeDma.D2H, 1 << 40, 10_000, 100, 0)                      # device -> file: destination is the mapping
        assert calls == [("copy", base + 10_000, 1 << 40, 100, FakeDma.D2H)]
        # releasing A frees windows 0 and 1; window 2 stays for B
        calls.clear()
        m.release(10_000, 140_000)
        assert calls == [("unreg", base), ("unreg", base + win)]
        calls.clear()
        m.release(150_000, 150_000)
        assert calls == [("unreg", base + 2 * win), ("unreg", base + 3 * win), ("unreg", base + 4 * win)] and m.ref == {}
        # the last window is clipped to the mapping's page-rounded length
        calls.clear()
        assert m.acquire(5 * win + 10, 1000)
        assert calls == [("reg", base + 5 * win, m.maplen - 5 * win)] and m.maplen % 4096 == 0 and m.maplen >= size
        m.release(5 * win + 10, 1000)
        # a refused window: what this call had registered is rolled back, what another batch holds stays
        assert m.acquire(0, 10)
        calls.clear()
        refuse.add(base + 2 * win)
        assert not m.acquire(10, 2 * win + 10)                             # windows 0 (held), 1 (new), 2 (refused)
        assert calls == [("reg", base + win, win), ("unreg", base + win)] and m.ref == {0: 1}
        m.close()
        assert ("unreg", base) in calls and m.ref == {}
    finally:
        os.close(fd)


def test_process_frames_takes_process_video_keywords():
    """pythoncrt_amd.process_frames — the drop-in for the loop of process_video (ref:1037-1131) — accepts every keyword process_video itself
    takes (tests/golden/reference_cli.json holds the reference's own parameter list): the effect keywords by name with the CLI's defaults, the
    container / codec ones (input_path, crf, nvenc_preset, ...) swallowed, anything else refused; its leading parameters are the things the
    reference's loop already holds (frame iterator, writer call, output size, fps, total_frames)."""
    import inspect
    import pythoncrt_amd as pc
    from pythoncrt_amd import render
    sig = inspect.signature(pc.process_frames)
    names = list(sig.parameters)
    assert names[:6] == ["frame_iter", "write_frame", "out_w", "out_h", "fps_out", "total_frames"]
    ref = FX["process_video_kwargs"]["defaults"]
    io_keys = set(render._IO_KEYS)
    for k, v in ref.items():
        if k in io_keys:
            assert k not in sig.parameters
            continue
        p = sig.parameters[k]
        assert p.kind is inspect.Parameter.KEYWORD_ONLY, k
        if k == "text_pos":
            assert tuple(p.default) == tuple(v)
        elif k == "text_after":
            assert p.default is True                          # process_video's own default (ref:910); the CLI passes its flag, off by default (ref:1199)
        elif k != "progress_cb":
            assert p.default == v, (k, p.default, v)          # = what process_video receives when the CLI is run with no flags
    assert set(ref) - io_keys <= set(names)
    assert sig.parameters["io_keywords"].kind is inspect.Parameter.VAR_KEYWORD


def test_process_frames_fails_loudly_without_a_gpu():
    """No CPU fallback in the product path: without a ROCm device process_frames raises (and a misspelt keyword is refused before anything else)."""
    import pytest
    import torch
    import pythoncrt_amd as pc
    with pytest.raises(TypeError):
        pc.process_frames(iter([]), lambda a: None, 8, 8, 30, 1, scanlines=1.0)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no ROCm device"):
            pc.process_frames(iter([np.zeros((8, 8, 3), np.uint8)]), lambda a: None, 8, 8, 30, 1)


def test_iter_rgb24_is_the_raw_readers_iteration():
    """pythoncrt_amd.iter_rgb24 = FFmpegRawReader.iter_frames (ref:495-506) over an open stream: whole frames in order, a trailing partial frame
    dropped, short reads of a pipe reassembled."""
    import io
    import pythoncrt_amd as pc
    h, w, n = 5, 7, 4
    data = np.arange(n * h * w * 3, dtype=np.uint32).astype(np.uint8).tobytes()
    got = list(pc.iter_rgb24(io.BytesIO(data + b"\x01\x02\x03"), w, h))
    assert len(got) == n and all(g.shape == (h, w, 3) and g.dtype == np.uint8 for g in got)
    assert b"".join(g.tobytes() for g in got) == data

    class Dribble:                                      # a pipe that hands out at most 11 bytes per read
        def __init__(self, b):
            self.b, self.p = b, 0

        def read(self, k):
            out = self.b[self.p:self.p + min(k, 11)]
            self.p += len(out)
            return out
    got2 = list(pc.iter_rgb24(Dribble(data), w, h))
    assert len(got2) == n and b"".join(g.tobytes() for g in got2) == data
    assert list(pc.iter_rgb24(io.BytesIO(b""), w, h)) == []
