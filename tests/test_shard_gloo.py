"""Frame-sharded render (pythoncrt_amd/shard.py) over gloo on CPU at world sizes 2, 3 and 8: the
schedule, the one-frame persistence carry and the p^j correction, with the CPU oracle plugged in
as the per-frame engine.  The sharded result must equal the single-process in-order render
(crt_filter.py ref:1081-1105) to float rounding.

World 2 cannot tell a ring's two directions apart ((r+1) % 2 == (r-1) % 2); worlds 3 and 8 (the target:
BASELINE configs[3] is an 8-GPU shard) have a distinct predecessor and successor, partial last rounds with
2 < active < world, rank 0 seeded from rank world-1, and an exact chain with several intermediate hops.
Every case of one (world, schedule) pair runs inside ONE process group — a spawn of 8 interpreters costs more
than the cases themselves."""
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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def in_order_render(p, n_frames):
    """The reference's loop: one process, frames committed strictly in order (ref:1081-1105)."""
    state, exp = None, []
    for i, f in enumerate(clip_frames(n_frames)):
        state, u8 = orc.persistence_blend(state, static_of(f, i), p)
        exp.append(u8)
    return np.stack(exp)


def cases_worker(rank, world, port, overlap, cases, outdir):
    """Every (p, chunk, n_frames) case through ShardedRender on this rank; the synchronous protocol (run_round) or the
    overlapped one (submit_round returns the PREVIOUS round's frames, flush() the last)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for k, (p, chunk, n_frames) in enumerate(cases):
        frames = clip_frames(n_frames)
        shard = FrameShard(world, rank, chunk)
        render = ShardedRender(shard, p, OracleEngine(p), dist=dist, overlap=overlap)
        cdir = os.path.join(outdir, f"case{k}")
        os.makedirs(cdir, exist_ok=True)
        seen = []

        def keep(done):
            for r, out in done:
                lo, _ = shard.frame_range(r, n_frames)
                seen.append(r)
                np.save(os.path.join(cdir, f"out_{lo}.npy"), out.numpy())

        if overlap:
            assert render.overlap == ((p ** chunk) < 2.0 ** -24 and p > 0.0 and world > 1)
        for r in range(shard.rounds(n_frames)):
            lo, hi = shard.frame_range(r, n_frames)
            mine = torch.from_numpy(np.stack(frames[lo:hi])) if hi > lo else None
            active = shard.active_ranks(r, n_frames)
            assert (mine is not None) == (rank < active)
            if overlap:
                done = render.submit_round(mine, r, active=active)
                if render.overlap and mine is not None:
                    assert all(rr < r for rr, _ in done)          # results arrive one call late
                keep(done)
            else:
                out = render.run_round(mine, r, active=active)
                assert (out is None) == (hi == lo)
                if out is not None:
                    keep([(r, out)])
        if overlap:
            keep(render.close())           # flush() + the staged schedule's worker thread and extra gloo groups released (collective)
            assert render.close() == []    # idempotent: nothing in flight, nothing left to free
        assert seen == sorted(seen) and len(seen) == len(shard.my_chunks(n_frames))
        dist.barrier()
    dist.destroy_process_group()


def check_cases(tmp_path, cases):
    for k, (p, chunk, n_frames) in enumerate(cases):
        got = np.concatenate([np.load(tmp_path / f"case{k}" / f"out_{lo}.npy") for lo in range(0, n_frames, chunk)])
        exp = in_order_render(p, n_frames)
        assert got.shape == exp.shape, (k, got.shape, exp.shape)
        d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 2e-3, (k, (p, chunk, n_frames), int(d.max()), float((d != 0).mean()))
        if p == 0.0:
            assert np.array_equal(got, exp)


def cases_for(world):
    """(p, chunk, n_frames).  p^chunk < 2^-24 -> one parallel hop per round, else the exact chain down the ring."""
    w = world
    return [
        (0.5, 26, 26 * w * 2),                       # parallel hop, full rounds only: rank 0 seeded from rank w-1
        (0.5, 26, 26 * w + 26 * (w // 2 + 1) + 7),   # ... + a partial last round: 2 < active < world at w = 8, short last chunk
        (0.3, 16, 16 * w * 2 + 16),                  # ... + a last round owned by rank 0 alone
        (0.5, 3, 3 * w * 3),                         # exact chain (p^B > 2^-24): w-1 hops per round, twice around the ring
        (0.9, 5, 5 * w + 5 * (w - 1) + 2),           # exact chain with a partial last round (active = w-1 >= 2) and a short chunk
        (0.9, 5, 5 * w + 5 * max(1, w // 2)),        # exact chain, partial last round of full chunks
        (0.0, 4, 4 * w + 9),                         # no persistence: no exchange at all, bit-equal
        (0.6, 4, 4),                                 # a clip of one chunk: active = 1
        (0.5, 31, 31 * w * 2),                       # parallel hop with a chunk that is not the settle length
    ]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_render_matches_in_order(tmp_path, world):
    """run_round at world 2 / 3 / 8: full rounds, ragged clips (partial last round, short last chunk), both hop
    protocols, p = 0."""
    cases = cases_for(world)
    mp.spawn(cases_worker, args=(world, free_port(), False, cases, str(tmp_path)), nprocs=world, join=True)
    check_cases(tmp_path, cases)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_overlapped_schedule_matches_in_order(tmp_path, world):
    """ShardedRender(overlap=True): round r's hop is in flight while round r+1 is scanned; the fix-up of round r runs
    one call later.  Same frames as the in-order render; schedules that cannot overlap (p = 0, exact ring chain) fall
    back to the synchronous protocol behind the same calls."""
    cases = cases_for(world)
    mp.spawn(cases_worker, args=(world, free_port(), True, cases, str(tmp_path)), nprocs=world, join=True)
    check_cases(tmp_path, cases)


def loopback_worker(rank, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=0, world_size=1)
    rows = []
    for overlap in (False, True):
        for p, chunk, n_frames in LOOPBACK_CASES:
            frames = clip_frames(n_frames)
            shard = FrameShard(1, 0, chunk)
            render = ShardedRender(shard, p, OracleEngine(p), dist=dist, overlap=overlap, loopback=True)
            outs = {}
            for r in range(shard.rounds(n_frames)):
                lo, hi = shard.frame_range(r, n_frames)
                for rr, o in render.submit_round(torch.from_numpy(np.stack(frames[lo:hi])), r, active=1):
                    outs[rr] = o.numpy()
            for rr, o in render.close():
                outs[rr] = o.numpy()
            np.save(os.path.join(outdir, f"lb_{int(overlap)}_{len(rows) % len(LOOPBACK_CASES)}.npy"), np.concatenate([outs[k] for k in sorted(outs)]))
            rows.append((render.overlap, render.parallel_hop))
    assert rows[:len(LOOPBACK_CASES)] == [(False, ph) for _, ph in rows[len(LOOPBACK_CASES):]]
    assert [ov for ov, _ in rows[len(LOOPBACK_CASES):]] == [ph for _, ph in rows[len(LOOPBACK_CASES):]]      # overlapped exactly where a round is one parallel hop
    dist.destroy_process_group()


LOOPBACK_CASES = [(0.5, 26, 26 * 3), (0.5, 26, 26 * 2 + 7), (0.5, 3, 10), (0.9, 5, 12), (0.0, 4, 9), (0.6, 4, 4)]


def test_loopback_ring_of_one_rank(tmp_path):
    """ShardedRender(loopback=True): world 1 run as a ring whose one rank is its own neighbour — the mode the RCCL branch is exercised with
    on a one-GPU box (tests/test_rccl_world1_gpu.py, bench.py --force-dist).  Same frames as the in-order render under both schedules and
    both hop protocols.  (gloo has no pair from a rank to itself, so the hop is a copy here; over RCCL it is a send / recv pair.)"""
    mp.spawn(loopback_worker, args=(free_port(), str(tmp_path)), nprocs=1, join=True)
    for overlap in (0, 1):
        for k, (p, chunk, n_frames) in enumerate(LOOPBACK_CASES):
            got = np.load(tmp_path / f"lb_{overlap}_{k}.npy")
            exp = in_order_render(p, n_frames)
            d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
            assert got.shape == exp.shape and d.max() <= 1 and (d != 0).mean() < 2e-3, (overlap, k, int(d.max()))
    with pytest.raises(ValueError):
        ShardedRender(FrameShard(2, 0, 26), 0.5, None, loopback=True)


def test_ring_direction_is_observable():
    """The protocol's peers at world 8 (what a world-2 test cannot see): rank r sends to r+1 and receives from r-1;
    in a partial round the ring is cut after the last active rank and rank 0 receives nothing."""
    class Probe(ShardedRender):
        def __init__(self, shard, p):
            super().__init__(shard, p, None)
            self.calls = []

        def _send_recv(self, send, recv_like, src, dst):
            self.calls.append((src, dst))
            return torch.zeros_like(recv_like)

    class Eng:
        def local_scan(self, frames, first, clip_start):
            n = frames.shape[0]
            return torch.zeros((n, 1, 1, 3)), torch.zeros((n, 1, 1, 3), dtype=torch.uint8)

        def correct(self, local, carry, p, out):
            pass

    fr = torch.zeros((26, 1, 1, 3), dtype=torch.uint8)
    for r in range(8):
        pr = Probe(FrameShard(8, r, 26), 0.5)
        pr.engine = Eng()
        pr.run_round(fr, 0)                        # full round
        assert pr.calls == [((r - 1) % 8, (r + 1) % 8)]
        pr.calls.clear()
        out = pr.run_round(fr if r < 5 else None, 1, active=5)      # partial round: ranks 0..4
        if r >= 5:
            assert out is None and pr.calls == []
        else:
            assert pr.calls == [(r - 1 if r > 0 else None, r + 1 if r + 1 < 5 else None)]
    # exact chain (p^B > 2^-24): true finals go down the ring one hop at a time, rank 7 closes it to rank 0 in a full round
    fr3 = torch.zeros((3, 1, 1, 3), dtype=torch.uint8)
    for r in range(8):
        pr = Probe(FrameShard(8, r, 3), 0.5)
        pr.engine = Eng()
        pr.run_round(fr3, 0)
        if r == 0:
            assert pr.calls == [(None, 1), (7, None)]
        else:
            assert pr.calls == [(r - 1, None), (None, (r + 1) % 8)]
        pr.calls.clear()
        pr.run_round(fr3 if r < 3 else None, 1, active=3)
        exp = {0: [(None, 1)], 1: [(0, None), (None, 2)], 2: [(1, None)]}.get(r, [])
        assert pr.calls == exp


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


def test_shard_plan_at_eight_ranks():
    """FrameShard at the target world size: every frame of ragged clips owned exactly once, in round-robin chunk order;
    the rounds / active-rank counts every rank derives agree; chunk c lands on rank c % 8 in round c // 8."""
    for chunk, n in [(512, 512 * 8 * 3), (512, 512 * 8 + 512 * 5 + 17), (26, 1), (26, 26 * 8 - 1), (1, 9), (7, 1000)]:
        shards = [FrameShard(8, r, chunk) for r in range(8)]
        rounds = {s.rounds(n) for s in shards}
        assert len(rounds) == 1
        n_rounds = rounds.pop()
        chunks = (n + chunk - 1) // chunk
        assert n_rounds == (chunks + 7) // 8
        owned = {}
        for s in shards:
            for i in range(n_rounds):
                lo, hi = s.frame_range(i, n)
                act = s.active_ranks(i, n)
                assert act == max(0, min(8, chunks - 8 * i))
                assert (hi > lo) == (s.rank < act)
                for t in range(lo, hi):
                    assert t not in owned and s.owner(t) == s.rank
                    owned[t] = (s.rank, i)
                if hi > lo:
                    assert lo // chunk == i * 8 + s.rank
        assert sorted(owned) == list(range(n))
        # only the clip's last chunk may be short
        for s in shards:
            for lo, hi in s.my_chunks(n):
                assert hi - lo == chunk or hi == n
