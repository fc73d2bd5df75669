"""Frame-sharded render (pythoncrt_amd/shard.py) with world_size 2 over gloo on CPU: the
schedule, the one-frame persistence carry and the p^j correction, with the CPU oracle plugged in
as the per-frame engine.  The sharded result must equal the single-process in-order render
(crt_filter.py ref:1081-1105) to float rounding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import crt_oracle as orc
from pythoncrt_amd.shard import FrameShard, ShardedRender, settle_frames

H, W = 24, 32
PARAMS = dict(scanline_strength=0.6, triad_gamma=2.2, triad_preserve_luma=False, aberration_px=1, bloom_sigma=0.0,
              bloom_strength=0.0, noise_strength=0.0, scanline_period_px=2.0, fast_bloom=False, pixel_size=1)
FPS, SPEED = 30.0, 30.0


def clip_frames(n):
    rng = np.random.default_rng(42)
    return [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(n)]


def static_of(frame, i):
    tm = orc.make_triad_mask(H, W, 0.35, 0.0)
    return orc.apply_static_effects(frame, PARAMS["scanline_strength"], tm, 2.2, False, 1, 0.0, 0.0, 0.0, 0.0, None, 2.0,
                                    (i / FPS) * SPEED, False, 1, 0, 0.0, time_sec=i / FPS)


class OracleEngine:
    """local_scan / correct of shard.ShardedRender on the CPU oracle."""

    def __init__(self, p):
        self.p = p

    def local_scan(self, frames, first_index, clip_start):
        state = None if clip_start else np.zeros((H, W, 3), np.float32)
        local, out = [], []
        for j in range(frames.shape[0]):
            st = static_of(frames[j].numpy(), first_index + j)
            state, u8 = orc.persistence_blend(state, st, self.p)
            local.append(np.asarray(state, np.float32))
            out.append(u8)
        return torch.from_numpy(np.stack(local)), torch.from_numpy(np.stack(out))

    def correct(self, local, carry, p, out):
        for j in range(local.shape[0]):
            v = np.clip(local[j].numpy() + np.float32(p ** (j + 1)) * carry.numpy(), 0.0, 1.0)
            out[j] = torch.from_numpy(orc.convert_scale_abs(v))


def worker(rank, world, port, p, chunk, rounds, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = clip_frames(chunk * world * rounds)
    shard = FrameShard(world, rank, chunk)
    render = ShardedRender(shard, p, OracleEngine(p), dist=dist)
    for r in range(rounds):
        lo, hi = shard.frame_range(r)
        mine = torch.from_numpy(np.stack(frames[lo:hi]))
        out = render.run_round(mine, r)
        np.save(os.path.join(outdir, f"out_{lo}.npy"), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def overlap_worker(rank, world, port, p, chunk, n_frames, outdir):
    """The overlapped schedule: submit_round returns the PREVIOUS round's frames, flush() the last."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = clip_frames(n_frames)
    shard = FrameShard(world, rank, chunk)
    render = ShardedRender(shard, p, OracleEngine(p), dist=dist, overlap=True)
    assert render.overlap == ((p ** chunk) < 2.0 ** -24 and p > 0.0)
    seen = []

    def keep(done):
        for r, out in done:
            lo, _ = shard.frame_range(r, n_frames)
            seen.append(r)
            np.save(os.path.join(outdir, f"out_{lo}.npy"), out.numpy())

    for r in range(shard.rounds(n_frames)):
        lo, hi = shard.frame_range(r, n_frames)
        mine = torch.from_numpy(np.stack(frames[lo:hi])) if hi > lo else None
        done = render.submit_round(mine, r, active=shard.active_ranks(r, n_frames))
        if render.overlap and mine is not None:
            assert all(rr < r for rr, _ in done)          # results arrive one call late
        keep(done)
    keep(render.flush())
    assert seen == sorted(seen) and len(seen) == len(shard.my_chunks(n_frames))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("p,chunk,n_frames", [(0.5, 26, 26 * 2 * 3), (0.5, 26, 26 * 5 + 7), (0.3, 16, 16 * 2 * 2 + 16), (0.9, 5, 25), (0.0, 4, 16)])
def test_two_rank_overlapped_schedule(tmp_path, p, chunk, n_frames):
    """ShardedRender(overlap=True): round r's hop is in flight while round r+1 is scanned; the fix-up of round r runs
    one call later.  Same frames as the in-order render; schedules that cannot overlap (p = 0, exact ring chain) fall
    back to the synchronous protocol behind the same calls."""
    world = 2
    mp.spawn(overlap_worker, args=(world, free_port(), p, chunk, n_frames, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"out_{lo}.npy") for lo in range(0, n_frames, chunk)])
    frames = clip_frames(n_frames)
    state, exp = None, []
    for i, f in enumerate(frames):
        state, u8 = orc.persistence_blend(state, static_of(f, i), p)
        exp.append(u8)
    exp = np.stack(exp)
    assert got.shape == exp.shape
    d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (int(d.max()), float((d != 0).mean()))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def ragged_worker(rank, world, port, p, chunk, n_frames, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = clip_frames(n_frames)
    shard = FrameShard(world, rank, chunk)
    render = ShardedRender(shard, p, OracleEngine(p), dist=dist)
    for r in range(shard.rounds(n_frames)):
        lo, hi = shard.frame_range(r, n_frames)
        mine = torch.from_numpy(np.stack(frames[lo:hi])) if hi > lo else None
        out = render.run_round(mine, r, active=shard.active_ranks(r, n_frames))
        assert (out is None) == (hi == lo)
        if out is not None:
            np.save(os.path.join(outdir, f"out_{lo}.npy"), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("p,chunk,n_frames", [(0.5, 3, 14), (0.5, 26, 60), (0.9, 5, 11), (0.0, 4, 9), (0.6, 4, 4)])
def test_two_rank_render_of_a_ragged_clip(tmp_path, p, chunk, n_frames):
    """The clip is not a multiple of world * chunk frames: the last round is owned by one rank only and / or its
    chunk is short.  Both the parallel-hop (p^B < 2^-24) and the exact ring chain are exercised."""
    world = 2
    mp.spawn(ragged_worker, args=(world, free_port(), p, chunk, n_frames, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"out_{lo}.npy") for lo in range(0, n_frames, chunk)])
    frames = clip_frames(n_frames)
    state, exp = None, []
    for i, f in enumerate(frames):
        state, u8 = orc.persistence_blend(state, static_of(f, i), p)
        exp.append(u8)
    exp = np.stack(exp)
    assert got.shape == exp.shape
    d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (int(d.max()), float((d != 0).mean()))


@pytest.mark.parametrize("p,chunk,rounds", [(0.5, 3, 3), (0.5, 26, 2), (0.5, 31, 2), (0.9, 5, 2), (0.0, 4, 2)])
def test_two_rank_render_matches_sequential(tmp_path, p, chunk, rounds):
    world = 2
    n = chunk * world * rounds
    mp.spawn(worker, args=(world, free_port(), p, chunk, rounds, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"out_{lo}.npy") for lo in range(0, n, chunk)])
    # single-process in-order render
    frames = clip_frames(n)
    state, exp = None, []
    for i, f in enumerate(frames):
        state, u8 = orc.persistence_blend(state, static_of(f, i), p)
        exp.append(u8)
    exp = np.stack(exp)
    d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (int(d.max()), float((d != 0).mean()))
    if p == 0.0:
        assert np.array_equal(got, exp)


def test_shard_plan():
    sh = FrameShard(8, 3, 16)
    assert sh.owner(0) == 0 and sh.owner(16 * 3) == 3 and sh.owner(16 * 11 + 5) == 3
    assert sh.frame_range(0) == (48, 64) and sh.frame_range(1) == (176, 192)
    assert sh.my_chunks(16 * 8 * 2) == [(48, 64), (176, 192)]
    assert sh.active_ranks(0, 300) == 8 and sh.active_ranks(2, 300) == 3 and sh.active_ranks(1, 16 * 8 * 2) == 8 and sh.rounds(300) == 3
    covered = sorted(t for r in range(8) for lo, hi in FrameShard(8, r, 16).my_chunks(300) for t in range(lo, hi))
    assert covered == list(range(300))
    assert settle_frames(0.5) == 24 and settle_frames(0.95) == 325 and settle_frames(0.0) == 0
    assert ShardedRender(FrameShard(2, 0, 26), 0.5, None).parallel_hop and not ShardedRender(FrameShard(2, 0, 3), 0.5, None).parallel_hop
    from pythoncrt_amd.shard import choose_chunk
    assert choose_chunk(0.5, 1920 * 1080 * 12, 128) == 128 and choose_chunk(0.95, 1920 * 1080 * 12, 128) == 325 and choose_chunk(0.0, 1, 32) == 32
    assert choose_chunk(0.95, 1 << 30, 16, mem_budget_bytes=64 << 30) == 32       # capped by memory: 2 slots x 32 x 1 GiB
