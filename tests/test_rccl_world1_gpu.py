"""SURVEY 8e on the hardware this pipeline has: the RCCL branch of the frame-sharded render at world size 1.

The reference's only parallel strategy is frames in parallel + strictly in-order persistence commit (crt_filter.py
ref:1015-1017, :1081-1105); shard.py maps it to one process per GPU with ONE float32 state frame hopping between ring
neighbours.  Until round 6 that hop had only ever run over gloo with host staging.  These tests run the real thing —
init_process_group("nccl"), a device-tensor isend / irecv pair, r.wait() ordering libcrtfx's stream behind RCCL's —
in a FRESH child process (RANK=0 WORLD_SIZE=1) on the box's one GPU, and bench.py's N > 1 code path the same way."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("CRTFX_DIST_BACKEND", None)
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-3000:]
    return json.loads(lines[0])


def test_rccl_self_hop_orders_the_fixup_and_matches_the_in_order_render():
    """1080p, chunks of 26 frames (settle_frames(0.5, 2^-26): every kept local state is corrected), three rounds."""
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_world1_child.py"), "1080", "1920", "26", "3"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    res = _json_line(r.stdout)
    assert res["ok"] and res["backend"] == "nccl" and res["backend_version"].startswith("rccl ") and res["world_size_seen"] == 1
    assert res["collectives_ok"] and res["self_hop_equal"]
    for key in ("hand_chunk0", "hand_chunk1"):
        assert res[key][0] <= 1 and res[key][1] < 2e-3, (key, res[key])
    assert res["uncorrected_chunk1"][1] > 0.01
    for key in ("ring_synchronous", "ring_overlapped", "ring_exact_chain"):
        assert res[key]["diff"][0] <= 1 and res[key]["diff"][1] < 2e-3, (key, res[key])
        assert res[key]["schedule"]["rounds"] >= 3
    assert res["ring_overlapped"]["overlap"] and not res["ring_synchronous"]["overlap"]
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_world1_child.json"), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("config,batch", [(4, 64), (3, 8)])
def test_bench_force_dist_runs_the_n_gt_1_path_over_rccl(config, batch):
    """bench.py --force-dist at world 1: init_process_group over RCCL, the barrier inside sync(), all_reduce(MAX) of the region time,
    all_gather / all_gather_object of the per-rank records; config 4 also runs the sharded-persistence schedule as the one-rank ring
    (zero-state scan, self hop, fix-up) and reports it."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--config", str(config),
           "--batch", str(batch), "--cpu-frames", "0", "--repeats", "0"]
    r = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    res = _json_line(r.stdout)
    # ONE JSON line and nothing else on stdout: RCCL's start-up banner ("RCCL version : ...", five lines) goes to stderr (bench.stdout_to_stderr)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[:600]
    d = res["dist"]
    assert res["n_gpus"] == 1 and res["value"] > 0
    assert d["backend"] == "nccl" and d["backend_version"].startswith("rccl ") and d["world_size_seen"] == 1 and d["forced"] is True
    assert d["per_rank_frames_per_s"] and len(d["per_rank"]) == 1 and d["per_rank"][0]["rank"] == 0
    if config == 4:
        assert d["hop_schedule"] == "synchronous"
        sr = res["shard_schedule"]
        assert sr["parallel_hop"] and sr["rounds"] >= 3 and sr["fixup_frames"] == 26 and sr["hop_stall_us"] >= 0
    else:
        assert d["hop_schedule"] is None


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_sharded_cli_over_rccl_at_world_1(tmp_path, overlap):
    """pythoncrt_amd.cli's sharded main (cli.py: eager RCCL communicator, broadcast_object_list of the grain seed, barriers, the hop between
    its upload / compute / download streams) as the one-rank ring (CRTFX_FORCE_DIST=1), both hop schedules, every download held back by 30 ms
    of GPU time: the file must hold the plain single-process render's frames (<= 1 LSB with the p^j carry correction, ragged last chunk)."""
    import numpy as np
    h, w, n_frames = 270, 480, 26 * 3 + 5
    rng = np.random.default_rng(23)
    frames = rng.integers(0, 256, (n_frames, h, w, 3), dtype=np.uint8)
    src = tmp_path / "in.rgb"
    src.write_bytes(frames.tobytes())
    flags = ["--width", str(w), "--height", str(h), "--fps", "30", "--batch", "26", "--noise-seed", "7", "--persistence", "0.5",
             "--no-fast-bloom", "--bloom-sigma", "1.2", "--warp-strength", "0.15", "--pixel-size", "1"]
    env = _env()
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    one, two = tmp_path / "one.rgb", tmp_path / "two.rgb"
    r = subprocess.run([sys.executable, "-m", "pythoncrt_amd.cli", "--input", str(src), "--output", str(one)] + flags,
                       env={k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    env.update(CRTFX_FORCE_DIST="1", CRTFX_SHARD_OVERLAP=overlap, CRTFX_TEST_DOWNLOAD_DELAY_MS="30")
    r = subprocess.run([sys.executable, "-m", "pythoncrt_amd.cli", "--input", str(src), "--output", str(two)] + flags,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"rank 0: {n_frames} of {n_frames} frames" in r.stderr
    a = np.frombuffer(one.read_bytes(), dtype=np.uint8)
    b = np.frombuffer(two.read_bytes(), dtype=np.uint8)
    assert a.size == b.size == frames.size
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (int(d.max()), float((d != 0).mean()))
