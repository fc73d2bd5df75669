"""GPU, SURVEY 8f rows 1 and 4: the CLI end to end on raw rgb24 files (pinned double-buffered staging, batches
that do not divide the clip, persistence carried across batches, the --text overlay), and an overlay whose
size differs from the frame's against the reference's own outputs (tests/golden/reference_text_overlay.npz)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import crt_oracle as orc  # noqa: E402  (checker only)

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pc():
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    import pythoncrt_amd
    return pythoncrt_amd


def clip(n, h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255) // (w - 1), (yy * 255) // (h - 1), ((xx + yy) * 255) // (h + w - 2)], axis=2)
    return np.stack([np.clip((base + 9 * i + rng.integers(0, 80, (h, w, 3))) // 2 + 30, 0, 255) for i in range(n)]).astype(np.uint8)


def run_cli(pc, tmp_path, frames, extra, batch):
    from pythoncrt_amd import cli
    n, h, w = frames.shape[:3]
    src, dst = tmp_path / "in.rgb", tmp_path / "out.rgb"
    src.write_bytes(frames.tobytes() + b"\x01\x02\x03")          # a trailing partial frame is ignored
    rc = cli.main(["--input", str(src), "--output", str(dst), "--width", str(w), "--height", str(h), "--fps", "30",
                   "--batch", str(batch), "--noise-seed", "99"] + extra)
    assert rc == 0
    out = np.frombuffer(dst.read_bytes(), dtype=np.uint8)
    assert out.size == n * h * w * 3
    return out.reshape(n, h, w, 3)


def test_cli_matches_pipeline_and_oracle(pc, tmp_path):
    """Reference CLI defaults (fast bloom, pixel_size 2, persistence 0.2, triad softness 0.5) with the grain off
    so that the oracle's process_frames can check the bytes; batch 3 over 8 frames = three batches, last short."""
    from pythoncrt_amd import cli
    from pythoncrt_amd.pipeline import FramePipeline
    n, h, w = 8, 72, 128
    frames = clip(n, h, w, 5)
    extra = ["--noise-strength", "0", "--warp-strength", "0.15"]
    out = run_cli(pc, tmp_path, frames, extra, batch=3)
    a = cli.build_parser().parse_args(["--input", "x"] + extra)
    rs = cli.settings_from_args(a)
    # one-shot pipeline run: same bytes whatever the batching
    dev = torch.device("cuda", 0)
    pipe = FramePipeline(dev, h, w, rs, fps=30, noise_seed=99)
    direct, _ = pipe.run(torch.from_numpy(frames).to(dev))
    assert np.array_equal(out, direct.cpu().numpy())
    # oracle (render loop ref:1037-1131)
    exp, _ = orc.process_frames(list(frames), dict(rs.__dict__), 30.0, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength,
                                rs.triad_softness, rs.vignette_strength)
    exp = np.stack(exp)
    d = np.abs(out.astype(np.int16) - exp.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())


def test_cli_pipes_equal_files(pc, tmp_path):
    """`--input - --output -` (the drop-in between two ffmpeg processes): the reader thread takes the frames from stdin
    sequentially, the writer thread puts them on stdout in order — the same bytes as the file-to-file run (memory-mapped input, positional
    writes), over more batches than there are staging slots, a short last batch, a trailing partial frame, persistence carried across
    batches; `--staging-report` goes to stderr only."""
    import subprocess
    import sys
    n, h, w = 11, 48, 96
    frames = clip(n, h, w, 8)
    extra = ["--noise-strength", "0.8", "--warp-strength", "0.1", "--persistence", "0.3"]
    ref = run_cli(pc, tmp_path, frames, extra, batch=2)
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "pythoncrt_amd.cli", "--input", "-", "--output", "-", "--width", str(w), "--height", str(h), "--fps", "30",
           "--batch", "2", "--noise-seed", "99", "--staging-report"] + extra
    r = subprocess.run(cmd, input=frames.tobytes() + b"\x07\x08", capture_output=True, cwd=root, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    out = np.frombuffer(r.stdout, dtype=np.uint8)
    assert out.size == n * h * w * 3, (out.size, r.stderr[-500:])
    assert np.array_equal(out.reshape(n, h, w, 3), ref)
    assert b"staging" in r.stderr and f"{n} frames".encode() in r.stderr


def test_cli_text_overlay_and_grain(pc, tmp_path):
    """--text draws through Pillow and blends on the GPU; the grain is reproducible from --noise-seed."""
    from pythoncrt_amd import cli, text
    from pythoncrt_amd.pipeline import FramePipeline
    n, h, w = 5, 64, 160
    frames = clip(n, h, w, 6)
    extra = ["--text", "CRT", "--text-size", "20", "--text-x", "8", "--text-y", "6", "--text-color", "#40FF80", "--no-fast-bloom",
             "--pixel-size", "1", "--persistence", "0"]
    out_before = run_cli(pc, tmp_path, frames, extra, batch=2)
    out_again = run_cli(pc, tmp_path, frames, extra, batch=4)
    assert np.array_equal(out_before, out_again)
    out_after = run_cli(pc, tmp_path, frames, extra + ["--text-after"], batch=2)
    plain = run_cli(pc, tmp_path, frames, extra[10:], batch=2)
    ov = text.make_text_overlay_rgba(w, h, "CRT", "", 20, "#40FF80", (8, 6))
    inked = ov[..., 3] > 0
    assert inked.any()
    # --text-after: pixels the overlay does not touch are the plain render's; fully opaque ones carry the ink colour
    assert np.array_equal(out_after[:, ~inked], plain[:, ~inked])
    solid = ov[..., 3] == 255
    if solid.any():
        assert np.array_equal(out_after[0][solid], ov[solid][:, :3])
    assert not np.array_equal(out_before, out_after)
    # against the pipeline called directly with the same overlay
    a = cli.build_parser().parse_args(["--input", "x"] + extra)
    pipe = FramePipeline(torch.device("cuda", 0), h, w, cli.settings_from_args(a), fps=30, noise_seed=99, text_overlay_rgba=ov,
                         text_overlay_after=False)
    direct, _ = pipe.run(torch.from_numpy(frames).cuda())
    assert np.array_equal(out_before, direct.cpu().numpy())


def test_overlay_of_another_size_matches_reference(pc):
    tg = np.load(os.path.join(HERE, "golden", "reference_text_overlay.npz"))
    ov, frame = tg["fit/overlay"], tg["fit/frame"]
    for after in (False, True):
        got = pc.apply_static_effects(frame, 0.0, None, 2.2, False, 0, 0.0, 0.0, 0.0, 0.0, None, 2.0, 0.0, False, 1, 0, 0.0,
                                      text_overlay_rgba=ov, text_overlay_after=after)
        exp = tg[f"fit/static_after{int(after)}"]
        assert got.shape == exp.shape and np.array_equal(got, exp.astype(np.float32))


@pytest.mark.parametrize("where", ["tmp", "shm"])
def test_cli_io_modes_write_the_same_bytes(pc, tmp_path, where, capsys):
    """--io staged (pinned slots + memcpy / pwrite), mapped (upload and download DMA straight from / into the registered file mappings) and auto
    (mapped where the first batch shows it pays): identical output files.  Frames of 27 648 bytes in batches of 3 — no batch boundary on a
    page, so neighbouring batches share a registered window — a short last batch, a clip shorter than one batch, a trailing partial frame,
    persistence carried across batches; on the test's own directory and (when the box has one) on tmpfs, where registration is the slow
    path and `auto` goes back to staging after the first batch."""
    import shutil
    import tempfile
    from pythoncrt_amd import cli
    base = tmp_path
    if where == "shm":
        if not os.path.isdir("/dev/shm") or not os.access("/dev/shm", os.W_OK):
            pytest.skip("no writable /dev/shm")
        base = type(tmp_path)(tempfile.mkdtemp(prefix="crtfx_io_", dir="/dev/shm"))
    try:
        for n, batch, (h, w) in ((11, 3, (72, 128)), (2, 5, (72, 128)), (6, 3, (72, 128)), (5, 2, (1080, 1920 * 4))):      # the last: 49.8 MB batches, several 32 MiB windows each
            frames = clip(n, h, w, 31 + n)
            src = base / "in.rgb"
            src.write_bytes(frames.tobytes() + b"\x01\x02")
            outs = {}
            for io in ("staged", "mapped", "auto"):
                dst = base / f"out_{io}.rgb"
                rc = cli.main(["--input", str(src), "--output", str(dst), "--width", str(w), "--height", str(h), "--fps", "30", "--batch", str(batch),
                               "--noise-seed", "5", "--warp-strength", "0.15", "--persistence", "0.4", "--io", io, "--staging-report"])
                assert rc == 0
                outs[io] = dst.read_bytes()
                err = capsys.readouterr().err
                line = [ln for ln in err.splitlines() if ln.startswith("staging: input")][0]
                if io == "staged":
                    assert "input 0 batches mapped" in line and "output 0 batches mapped" in line, line
                elif io == "mapped" and "refused" not in line:
                    # (the reader always asks for one batch more than the clip's whole batches: the short — possibly empty — one that ends it)
                    assert f"input {n // batch + 1} batches mapped" in line and f"output {-(-n // batch)} batches mapped" in line, line
            assert len(outs["staged"]) == frames.size
            assert outs["mapped"] == outs["staged"] and outs["auto"] == outs["staged"], (n, batch)
    finally:
        if where == "shm":
            shutil.rmtree(str(base), ignore_errors=True)


@pytest.mark.parametrize("persistence,n_frames", [("0.5", 11), ("0", 9)])
def test_sharded_cli_two_ranks(pc, tmp_path, persistence, n_frames):
    """SURVEY 8e end to end: two ranks (one process each, launched by torch.distributed.run; both on this box's single
    GPU, so the state frame is exchanged through gloo's host staging instead of RCCL) render a ragged clip from a raw
    file into a raw file; the bytes must be the single-process render's (exactly without persistence, <= 1 LSB with
    the p^j carry correction)."""
    import socket
    import subprocess
    import sys
    from pythoncrt_amd import cli
    h, w = 72, 128
    frames = clip(n_frames, h, w, 9)
    src = tmp_path / "in.rgb"
    src.write_bytes(frames.tobytes())
    flags = ["--width", str(w), "--height", str(h), "--fps", "30", "--batch", "3", "--noise-seed", "7", "--persistence", persistence,
             "--no-fast-bloom", "--bloom-sigma", "1.2", "--warp-strength", "0.15", "--pixel-size", "1"]
    one = tmp_path / "one.rgb"
    assert cli.main(["--input", str(src), "--output", str(one)] + flags) == 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = tmp_path / "two.rgb"
    env = dict(os.environ, CRTFX_DIST_BACKEND="gloo", PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "pythoncrt_amd.cli", "--input", str(src), "--output", str(two)] + flags
    r = subprocess.run(cmd, env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    a = np.frombuffer(one.read_bytes(), dtype=np.uint8)
    b = np.frombuffer(two.read_bytes(), dtype=np.uint8)
    assert a.size == b.size == frames.size
    if persistence == "0":
        assert np.array_equal(a, b)
    else:
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 2e-3, (int(d.max()), float((d != 0).mean()))


@pytest.mark.parametrize("persistence", ["0", "0.5"])
def test_sharded_cli_synchronous_schedule(pc, tmp_path, persistence):
    """The schedule the sharded CLI runs over RCCL by default, and for every render without persistence: ShardedRender.run_round, one round
    finished per call, its frames downloaded asynchronously while the next round is scanned.  Every round must write its own output slot
    (round 4's advisor finding: run_round always wrote slot 0, so round r + 1's scan could overwrite frames round r's download was still
    reading, and the small-frame test passed by timing).  Here every download is held back by 30 ms of GPU time
    (CRTFX_TEST_DOWNLOAD_DELAY_MS): a missing slot rotation or ordering gives wrong bytes every time."""
    import socket
    import subprocess
    import sys
    from pythoncrt_amd import cli
    h, w, n_frames = 72, 128, 17
    frames = clip(n_frames, h, w, 19)
    src = tmp_path / "in.rgb"
    src.write_bytes(frames.tobytes())
    flags = ["--width", str(w), "--height", str(h), "--fps", "30", "--batch", "2", "--noise-seed", "7", "--persistence", persistence,
             "--no-fast-bloom", "--bloom-sigma", "1.2", "--warp-strength", "0.15", "--pixel-size", "1"]
    one = tmp_path / "one.rgb"
    assert cli.main(["--input", str(src), "--output", str(one)] + flags) == 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = tmp_path / "two.rgb"
    env = dict(os.environ, CRTFX_DIST_BACKEND="gloo", CRTFX_SHARD_OVERLAP="0", CRTFX_TEST_DOWNLOAD_DELAY_MS="30",
               PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "pythoncrt_amd.cli", "--input", str(src), "--output", str(two)] + flags
    r = subprocess.run(cmd, env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    a = np.frombuffer(one.read_bytes(), dtype=np.uint8)
    b = np.frombuffer(two.read_bytes(), dtype=np.uint8)
    assert a.size == b.size == frames.size
    if persistence == "0":
        assert np.array_equal(a, b)
    else:
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 2e-3, (int(d.max()), float((d != 0).mean()))


def test_c_abi_consumer(pc, tmp_path):
    """examples/crtfx_c_abi.cpp: a program that uses only include/crtfx.h, libcrtfx.so and the HIP runtime (no Python
    or torch in its call path; it builds the vignette and warp axis tables itself) renders the same bytes as the Python
    host with the same settings."""
    import shutil
    import subprocess
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    root = os.path.dirname(HERE)
    exe = os.path.join(root, "build", "crtfx_c_abi")
    src = os.path.join(root, "examples", "crtfx_c_abi.cpp")
    lib = os.path.join(root, "pythoncrt_amd", "libcrtfx.so")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-I" + os.path.join(root, "include"), src, "-L" + os.path.dirname(lib), "-lcrtfx",
                        "-o", exe], check=True, timeout=300)
    n, h, w = 5, 72, 128
    frames = clip(n, h, w, 11)
    fin, fout = tmp_path / "in.rgb", tmp_path / "out.rgb"
    fin.write_bytes(frames.tobytes())
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(lib) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(fin), str(w), str(h), str(n), str(fout)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "frames of 128x72" in r.stdout
    assert "plan: point=k_point" in r.stdout and ";warp=k_warp" in r.stdout, r.stdout          # crtfx_last_plan from plain C
    got = np.frombuffer(fout.read_bytes(), dtype=np.uint8).reshape(n, h, w, 3)
    rs = RenderSettings(scanline_strength=0.0, triad_strength=0.0, aberration_px=2, bloom_strength=0.0, noise_strength=0.0,
                        vignette_strength=0.4, persistence=0.5, fast_bloom=False, pixel_size=1, warp_strength=0.2)
    pipe = FramePipeline(torch.device("cuda", 0), h, w, rs, fps=30.0, noise_seed=0)
    exp, _ = pipe.run(torch.from_numpy(frames).cuda())
    assert np.array_equal(got, exp.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("config,batch", [(2, 8), (4, 26)])
def test_bench_launches_its_own_ranks(config, batch):
    """`python bench.py --gpus 2` as the driver calls it (no torch.distributed.run around it): the parent starts the two
    ranks as child processes and relays rank 0's JSON line.  Both ranks share the test box's one GPU (gloo rehearsal);
    config 4 exercises the overlapped sharded-persistence schedule (chunk 26 >= settle_frames(0.5) = 24)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    env = dict(os.environ, CRTFX_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", str(config),
           "--batch", str(batch), "--cpu-frames", "0", "--repeats", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["dist"]["world_size_seen"] == 2 and len(res["dist"]["per_rank_frames_per_s"]) == 2
    assert res["value"] > 0 and res["config"]["frames_per_step_per_gpu"] == batch
    # every rank's own view of the reported region: frames/s on its own clock, its kernel time per frame, its GPU's clock / power samples
    pr = res["dist"]["per_rank"]
    assert [d["rank"] for d in pr] == [0, 1]
    for d in pr:
        assert d["frames_per_s_own_clock"] > 0 and d["chain_ms_per_frame"] > 0 and set(d["gpu"]) >= {"samples", "sclk_mhz_mean", "power_w_mean"}
    assert res["dist"]["backend"] == "gloo" and res["dist"]["backend_version"].startswith("gloo") and res["dist"]["visible_devices"] >= 1
    assert res["dist"]["hop_schedule"] == ("overlapped" if config == 4 else None)
    if config == 4:
        sr = res["shard_schedule"]
        assert sr["overlap"] and sr["parallel_hop"] and sr["rounds"] >= 2 and sr["fixup_frames"] == 26
        assert all(d["shard_schedule"]["rounds"] == sr["rounds"] for d in pr)      # the schedule of EVERY rank, not rank 0's only


@pytest.mark.parametrize("case", ["too_many_frames", "more_ranks_than_gpus"])
def test_bench_preflight_fails_fast_with_a_reason(case):
    """An N-rank bench run that cannot work ends in seconds, non-zero, with ONE line saying why — before the rendezvous, before any large
    allocation: (a) a batch that does not fit the device memory; (b) --gpus 2 over RCCL on a box with fewer devices (when this box has fewer
    than 2).  (The rendezvous itself carries a 60 s timeout for a rank that never arrives.)"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    if case == "too_many_frames":
        env["CRTFX_DIST_BACKEND"] = "gloo"
        extra, want = ["--batch", "400000"], "lower --batch"
    else:
        if torch.cuda.device_count() >= 2:
            pytest.skip("this box has two devices: the launch is valid")
        env.pop("CRTFX_DIST_BACKEND", None)
        extra, want = ["--batch", "8"], "visible devices"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "3", "--cpu-frames", "0", "--repeats", "0"] + extra
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0 and took < 60, (r.returncode, took, r.stderr[-1500:])
    assert "bench.py preflight failed" in r.stderr and want in r.stderr, r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # no result line


@pytest.mark.parametrize("case", ["defaults", "gaussian_warp_text", "no_persistence_resized"])
def test_process_frames_is_the_render_loop(pc, case):
    """pythoncrt_amd.process_frames — the loop of process_video (ref:1037-1131) over the caller's own frame iterator and writer call: frames
    in order, frame i at phase i / fps * speed, persistence carried across its batches, a frame of another size resized with Pillow first
    (ref:1039-1041), progress_cb after every frame (ref:1104-1105), text overlay built once.  Same bytes as one FramePipeline.run over the whole
    clip, and the oracle's in-order render within the usual bar."""
    from pythoncrt_amd.pipeline import FramePipeline, RenderSettings
    from pythoncrt_amd.text import make_text_overlay_rgba
    n, h, w, fps = 11, 72, 128, 25
    frames = list(clip(n, h, w, 77))
    kw = {"defaults": dict(noise_strength=0.0),
          "gaussian_warp_text": dict(fast_bloom=False, bloom_sigma=2.0, pixel_size=1, warp_strength=0.15, persistence=0.5, noise_strength=0.0,
                                     text="CRT", text_size=20, text_pos=(8, 8), text_after=True, text_color="#FFCC00"),
          "no_persistence_resized": dict(persistence=0.0, noise_strength=0.0, fast_bloom=False, bloom_sigma=1.2, pixel_size=1)}[case]
    feed = list(frames)
    if case == "no_persistence_resized":
        from PIL import Image
        big = np.asarray(Image.fromarray(frames[4]).resize((w * 2, h * 2), Image.BILINEAR))      # a frame of another size in the stream
        feed[4] = big
        frames[4] = np.asarray(Image.fromarray(big).resize((w, h), Image.BILINEAR))              # what the loop makes of it (ref:1039-1041)
    got, prog = [], []
    written = pc.process_frames(iter(feed), lambda a: got.append(np.array(a)), w, h, fps, n, batch=4, noise_seed=5, progress_cb=prog.append,
                                crf=18, nvenc_preset="p4", input_path="x.mp4", **kw)
    assert written == n and len(got) == n and all(g.shape == (h, w, 3) and g.dtype == np.uint8 for g in got)
    assert prog == [min(1.0, (i + 1) / n) for i in range(n)]
    # the same clip in one piece
    fx = {k: v for k, v in kw.items() if not k.startswith("text")}
    rs = RenderSettings(**fx)
    dev = torch.device("cuda", torch.cuda.current_device())
    ov = make_text_overlay_rgba(w, h, kw["text"], "", kw["text_size"], kw["text_color"], kw["text_pos"]) if "text" in kw else None
    pipe = FramePipeline(dev, h, w, rs, fps=fps, noise_seed=5, text_overlay_rgba=ov, text_overlay_after=kw.get("text_after", True))
    direct, _ = pipe.run(torch.from_numpy(np.stack(frames)).to(dev))
    assert np.array_equal(np.stack(got), direct.cpu().numpy())
    # ... and the oracle's in-order loop
    params = {k: getattr(rs, k) for k in ("scanline_strength", "triad_gamma", "triad_preserve_luma", "aberration_px", "bloom_sigma", "bloom_strength",
                                          "bloom_threshold", "noise_strength", "scanline_period_px", "fast_bloom", "pixel_size", "warp_strength")}
    if ov is not None:
        params.update(text_overlay_rgba=ov, text_overlay_after=True)
    exp, _ = orc.process_frames(frames, params, fps, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness, rs.vignette_strength)
    d = np.abs(np.stack(got).astype(np.int16) - np.stack(exp).astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (case, int(d.max()), float((d != 0).mean()))
    # a keyword that is neither an effect nor one of process_video's I/O keywords is refused
    with pytest.raises(TypeError):
        pc.process_frames(iter(feed), lambda a: None, w, h, fps, n, scanlines=0.5)


@pytest.mark.parametrize("io", ["staged", "mapped"])
def test_cli_failed_render_leaves_no_stale_frames(pc, tmp_path, monkeypatch, io):
    """Round-5 advisor finding: a file-to-file render over an EXISTING longer output that dies midway must not leave a full-length file whose
    tail holds frames of the earlier render.  The third batch's launch is made to raise; afterwards the file holds exactly the frames that were
    written (a prefix of the correct render, whole frames), for the growing staged output and for the pre-sized mapped one."""
    from pythoncrt_amd import cli
    from pythoncrt_amd.pipeline import FramePipeline
    n, h, w, batch = 14, 72, 128, 3
    frames = clip(n, h, w, 41)
    fb = h * w * 3
    src, dst = tmp_path / "in.rgb", tmp_path / "out.rgb"
    src.write_bytes(frames.tobytes())
    flags = ["--input", str(src), "--output", str(dst), "--width", str(w), "--height", str(h), "--fps", "30", "--batch", str(batch), "--noise-seed", "3",
             "--io", io]
    assert cli.main(flags) == 0
    good = dst.read_bytes()
    assert len(good) == n * fb
    dst.write_bytes(b"\xaa" * ((n + 6) * fb))                  # an earlier, longer render
    real, calls = FramePipeline.run, [0]

    def run(self, *a, **k):
        calls[0] += 1
        if calls[0] == 3:
            raise RuntimeError("injected failure in batch 3")
        return real(self, *a, **k)
    monkeypatch.setattr(FramePipeline, "run", run)
    with pytest.raises(RuntimeError, match="injected failure"):
        cli.main(flags)
    monkeypatch.undo()
    left = dst.read_bytes()
    assert len(left) % fb == 0 and len(left) <= 2 * batch * fb, (io, len(left) // fb)
    assert left == good[:len(left)]                            # whole frames of THIS render; nothing of the 0xAA file survives
    assert cli.main(flags) == 0 and dst.read_bytes() == good   # and the next render is unaffected


def test_process_frames_unknown_total_and_failing_writer(pc):
    """process_frames with total_frames=None: no fraction can be formed, so progress_cb fires once, with 1.0, after the last frame; a write_frame
    that raises ends the call with that exception after the queued GPU work has been drained (the next call on the same device works)."""
    n, h, w = 9, 72, 128
    frames = list(clip(n, h, w, 13))
    got, prog = [], []
    assert pc.process_frames(iter(frames), lambda a: got.append(np.array(a)), w, h, 25, None, batch=4, noise_seed=5, progress_cb=prog.append) == n
    assert prog == [1.0] and len(got) == n

    def bad(a):
        if len(seen) == 5:
            raise OSError("encoder pipe closed")
        seen.append(1)
    seen = []
    with pytest.raises(OSError, match="encoder pipe closed"):
        pc.process_frames(iter(frames), bad, w, h, 25, n, batch=4, noise_seed=5)
    again = []
    assert pc.process_frames(iter(frames), lambda a: again.append(np.array(a)), w, h, 25, n, batch=4, noise_seed=5) == n
    assert np.array_equal(np.stack(again), np.stack(got))
