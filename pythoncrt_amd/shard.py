"""Frame sharding across ranks — one process per GPU.

The reference parallelises over frames with a 2-thread pool and commits frames strictly in order
because of the persistence IIR  state_t = clip(p*state_{t-1} + (1-p)*static_t)  (crt_filter.py
ref:1015-1017, :1081-1105).  Static effects are independent per frame, so frames shard with no
communication at all when persistence is 0.  With persistence p > 0 the only coupling is that
first-order linear recurrence (its clip is inactive: both inputs lie in [0,1]), so

    state_t = local_t + p^(t - t0 + 1) * carry_in

where local_t is the scan of a chunk started from a ZERO incoming state at frame t0 and carry_in
is the true final state of the previous chunk.  Each rank therefore

  1. scans its own chunk locally (all the expensive work, fully parallel),
  2. exchanges ONE float32 state frame with its ring neighbours (RCCL send/recv over one xGMI
     link; gloo in the CPU tests) — never an all-reduce / all-gather,
  3. re-quantises its frames with the p^j-weighted carry added.

Chunks are dealt round-robin: chunk c (frames [c*B, (c+1)*B)) belongs to rank c % world and is
processed in round c // world.  The module is engine-agnostic (the GPU engine lives in
pipeline.py; the CPU tests plug the oracle in) and imports nothing GPU-specific.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch


@dataclass
class FrameShard:
    world: int
    rank: int
    chunk: int      # B frames per chunk

    def owner(self, t: int) -> int:
        return (t // self.chunk) % self.world

    def chunk_of(self, round_index: int) -> int:
        return round_index * self.world + self.rank

    def frame_range(self, round_index: int, n_frames: Optional[int] = None) -> Tuple[int, int]:
        c = self.chunk_of(round_index)
        lo, hi = c * self.chunk, (c + 1) * self.chunk
        if n_frames is not None:
            lo, hi = min(lo, n_frames), min(hi, n_frames)
        return lo, hi

    def rounds(self, n_frames: int) -> int:
        chunks = (n_frames + self.chunk - 1) // self.chunk
        return (chunks + self.world - 1) // self.world

    def active_ranks(self, round_index: int, n_frames: int) -> int:
        """Ranks that own a (possibly short) chunk in this round of an n_frames clip: world, except in a partial last round."""
        chunks = (n_frames + self.chunk - 1) // self.chunk
        return max(0, min(self.world, chunks - round_index * self.world))

    def my_chunks(self, n_frames: int) -> List[Tuple[int, int]]:
        return [r for r in (self.frame_range(i, n_frames) for i in range(self.rounds(n_frames))) if r[1] > r[0]]


def settle_frames(p: float, eps: float = 2.0 ** -24) -> int:
    """Frames after which a unit incoming state has decayed below eps: ceil(ln eps / ln p)."""
    if p <= 0.0:
        return 0
    return int(math.ceil(math.log(eps) / math.log(p)))


def choose_chunk(p: float, state_frame_bytes: int, default: int, mem_budget_bytes: int = 64 << 30, slots: int = 2) -> int:
    """Frames per chunk for a sharded render with persistence p: at least settle_frames(p), so that every round is ONE
    parallel hop instead of the world-1 hop chain (p = 0.5 -> 24, p = 0.95 -> 325), as long as the per-frame local
    states of `slots` chunks (double buffering) fit the memory budget — 288 GB of HBM per GPU is what makes this the
    default rather than a special case (1080p: 24.9 MB per state frame, 4K: 99.5 MB)."""
    if p <= 0.0:
        return int(default)
    want = max(int(default), settle_frames(p))
    cap = max(1, int(mem_budget_bytes // max(1, slots * state_frame_bytes)))
    return min(want, cap) if cap < want else want


class ShardedRender:
    """The per-round protocol.  `engine` provides

        local_scan(frames, first_index, clip_start[, slot]) -> (local_states float32 (n,H,W,3), out uint8 (n,H,W,3))
            scan of the chunk from a zero incoming state (clip_start: the very first frame of the
            clip passes through unblended, ref:1094-1095, and there is no carry at all); `slot` (0 .. engine.slots - 1) names the
            buffer set to use when the engine multi-buffers (engine.slots >= 2)
        correct(local_states, carry, p, out) -> None
            out[j] = quantise(clip(local_states[j] + p^(j+1) * carry))

    `dist` is torch.distributed (or None for world 1).  When p^B is below float32 epsilon the
    chunk-final LOCAL state already equals the true one to rounding, so every rank forwards it
    at once (one parallel hop per round); otherwise the true finals are forwarded down the ring
    rank by rank (exact for any B, at the cost of a world-1 hop chain).

    overlap=True (parallel-hop rounds only): `submit_round` enqueues round r's local scan and starts its hop, then
    finishes round r-1 (waits for ITS hop — long done — and runs its fix-up), so the state frame travels while the
    next round's scan runs; results come back one call late, `flush()` returns the last.  The overlapped schedule has
    run over gloo only (CPU tests at world 2 / 3 / 8, several ranks on one GPU); callers keep the synchronous run_round
    as the default over RCCL until it has run on a multi-GPU box (bench.py, cli.main_sharded: CRTFX_SHARD_OVERLAP=1 opts in).

    loopback=True (world 1 only): the ring of ONE rank.  Rank 0's successor and predecessor are both rank 0, so the
    protocol below runs unchanged — zero-state scan, the chunk-final state sent to and received from itself inside one
    batch_isend_irecv group, p^j fix-up — instead of the sequential-scan shortcut a single rank normally takes.  The
    frames are those of the in-order render (the chunk before chunk c IS this rank's previous round).  It exists so
    that the RCCL branch (device-tensor hop, r.wait() ordering the compute stream behind RCCL's stream, both
    schedules) can run and be checked on a one-GPU box: tests/test_rccl_world1_gpu.py, bench.py --force-dist."""

    def __init__(self, shard: FrameShard, persistence: float, engine, dist=None, group=None, overlap: bool = False, timing: bool = False,
                 loopback: bool = False):
        self.shard, self.p, self.engine, self.dist, self.group = shard, float(persistence), engine, dist, group
        if loopback and (shard.world != 1 or dist is None):
            raise ValueError("loopback=True is the one-rank ring: world 1 with an initialised process group")
        self.loopback = bool(loopback)
        self.carry_next_round: Optional[torch.Tensor] = None      # rank 0: true final of the previous round's last chunk
        self.parallel_hop = self.p > 0.0 and (self.p ** shard.chunk) < 2.0 ** -24
        self.overlap = bool(overlap) and self.parallel_hop and (shard.world > 1 or self.loopback)
        self.timing = bool(timing)
        self._worker = None           # staged (gloo + device tensors) overlapped hops run on one worker thread
        # gloo moves one message over ONE TCP pair (~4 GB/s of loopback memcpy): the staged hop of an overlapped schedule
        # splits the state frame over a few extra process groups = a few sockets and gloo threads in parallel.  (RCCL: unused.)
        self._channels = []
        if self.overlap and dist is not None and dist.get_backend(group) == "gloo":
            import os
            for _ in range(max(0, min(8, int(os.environ.get("CRTFX_GLOO_CHANNELS", "8"))) - 1)):
                self._channels.append(dist.new_group(backend="gloo"))      # collective: every rank constructs its ShardedRender
        self._last_transport_s = None
        self._pending = None          # overlapped mode: the round whose hop is in flight
        self._marks = []              # timing: per finished round (scan events, hop-wait events + host seconds, fix-up events); the last MARKS_KEPT rounds

    MARKS_KEPT = 256

    def _mark(self, m):
        self._marks.append(m)
        if len(self._marks) > self.MARKS_KEPT:
            del self._marks[:len(self._marks) - self.MARKS_KEPT]

    def close(self):
        """Finish the round in flight, stop the staging worker and destroy the extra gloo groups of a staged overlapped
        schedule (collective: every rank calls it, like the constructor).  Returns what flush() returns."""
        done = self.flush()
        if self._worker is not None:
            self._jobs.put(None)
            self._worker.join(timeout=30)
            self._worker = None
        for g in self._channels:
            self.dist.destroy_process_group(g)
        self._channels = []
        return done

    # ---- point-to-point plumbing --------------------------------------------------------------------------------
    def _staged(self, like: torch.Tensor) -> bool:
        # RCCL moves device tensors directly (one xGMI hop).  gloo has no device point-to-point: stage through the
        # host (CPU tests, and rehearsals of several ranks on one GPU).
        return like.is_cuda and self.dist.get_backend(self.group) == "gloo"

    def _post_staged_async(self, send: Optional[torch.Tensor], recv_like: torch.Tensor, src: Optional[int], dst: Optional[int]):
        """gloo rehearsal of a device hop, off the calling thread: the state frame is copied to pinned host memory on a
        side stream once the scan that produces it has finished (an event, not a host sync — the compute stream keeps
        its queue), then sent / received by ONE worker thread in round order.  Returns a handle for _complete."""
        import queue
        import threading
        if self._worker is None:
            self._jobs = queue.Queue()

            def run():
                while True:
                    job = self._jobs.get()
                    if job is None:
                        return
                    fn, box, done = job
                    try:
                        box.append(fn())
                    except BaseException as e:      # surfaced by _complete on the calling thread
                        box.append(e)
                    done.set()
            self._worker = threading.Thread(target=run, daemon=True)
            self._worker.start()
            self._side = torch.cuda.Stream(device=recv_like.device)
        ready = torch.cuda.Event()
        ready.record()                                   # behind the scan on the compute stream
        d, group, side = self.dist, self.group, self._side
        host_send = torch.empty(recv_like.shape, dtype=recv_like.dtype, pin_memory=True) if (dst is not None and send is not None) else None
        host_recv = torch.empty(recv_like.shape, dtype=recv_like.dtype, pin_memory=True) if src is not None else None

        chans = [group] + list(self._channels)

        def job():
            # slice k of the frame travels on channel k, and the three legs are pipelined slice by slice: device -> pinned host
            # on the side stream, the socket, pinned host -> device on the side stream again (one slice's copy under the next
            # slice's transfer: the chain costs the transfer + one slice's copies instead of the sum of the three)
            import time as _t
            t0 = _t.perf_counter()
            nch = len(chans)
            fs = host_send.view(-1) if host_send is not None else None
            fr = host_recv.view(-1) if host_recv is not None else None
            n_el = (fs if fs is not None else fr).numel() if (fs is not None or fr is not None) else 0
            step = (n_el + nch - 1) // nch if n_el else 0
            spans = [(k * step, min(n_el, (k + 1) * step)) for k in range(nch) if min(n_el, (k + 1) * step) > k * step]
            recvs = [d.irecv(fr[lo:hi], src, group=chans[k]) for k, (lo, hi) in enumerate(spans)] if fr is not None else []
            staged = []
            if fs is not None:
                flat_dev = send.reshape(-1)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    for lo, hi in spans:
                        fs[lo:hi].copy_(flat_dev[lo:hi], non_blocking=True)
                        e = torch.cuda.Event()
                        e.record(side)
                        staged.append(e)
            t1 = _t.perf_counter()
            sends = []
            for k, e in enumerate(staged):
                e.synchronize()
                if k == 0:
                    t1 = _t.perf_counter()              # the scan has finished and the first slice is on the host: the transfer starts
                lo, hi = spans[k]
                sends.append(d.isend(fs[lo:hi], dst, group=chans[k]))
            dev_recv, landed = None, None
            if fr is not None:
                with torch.cuda.stream(side):
                    dev_recv = torch.empty(recv_like.shape, dtype=recv_like.dtype, device=recv_like.device)
                flat_out = dev_recv.view(-1)
                for k, w in enumerate(recvs):
                    w.wait()
                    lo, hi = spans[k]
                    with torch.cuda.stream(side):
                        flat_out[lo:hi].copy_(fr[lo:hi], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(side)
            for w in sends:
                w.wait()
            return (dev_recv, landed, host_recv, host_send), _t.perf_counter() - t1, t1 - t0
        box, done = [], threading.Event()
        self._jobs.put((job, box, done))
        return ("async", box, done)

    def _post(self, send: Optional[torch.Tensor], recv_like: torch.Tensor, src: Optional[int], dst: Optional[int]):
        """Start the exchange; returns (works, recv buffer or None, staged)."""
        ops, recv = [], None
        d = self.dist
        if self.loopback and d.get_backend(self.group) == "gloo":
            # gloo has no pair from a rank to itself: the one-rank ring's hop is a copy there (CPU test of the protocol).  Over RCCL the
            # same call is a real ncclSend / ncclRecv pair inside one group — the branch the loopback mode exists to run.
            return [], (send.clone() if (send is not None and src is not None) else None), False
        stage = self._staged(recv_like)
        if dst is not None and send is not None:
            ops.append(d.P2POp(d.isend, send.cpu() if stage else send.contiguous(), dst, self.group))
        if src is not None:
            recv = torch.empty_like(recv_like, device="cpu") if stage else torch.empty_like(recv_like)
            ops.append(d.P2POp(d.irecv, recv, src, self.group))
        works = d.batch_isend_irecv(ops) if ops else []
        return works, recv, stage

    def _complete(self, works, recv, stage, device):
        if isinstance(works, str) and works == "async":      # (tag, box, done) from _post_staged_async
            box, done = recv, stage
            done.wait()
            res = box[0]
            if isinstance(res, BaseException):
                raise res
            (dev_recv, landed, _keep_r, _keep_s), transport_s, stage_s = res
            self._last_transport_s = transport_s
            if dev_recv is None:
                return None
            cur = torch.cuda.current_stream(device)
            cur.wait_event(landed)                      # the last slice has reached the device (side stream)
            dev_recv.record_stream(cur)
            return dev_recv
        for r in works:
            r.wait()            # RCCL: orders the current stream behind the transfer (no host block); gloo: host wait
        return recv.to(device) if (stage and recv is not None) else recv

    def _send_recv(self, send: Optional[torch.Tensor], recv_like: torch.Tensor, src: Optional[int], dst: Optional[int]):
        works, recv, stage = self._post(send, recv_like, src, dst)
        return self._complete(works, recv, stage, recv_like.device)

    # ---- timing (schedule_report) ---------------------------------------------------------------------------------
    def _ev(self, like: Optional[torch.Tensor]):
        if not self.timing or like is None or not like.is_cuda:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def schedule_report(self):
        """Mean per-round split of the sharded-persistence schedule on this rank: local scan, stall of the compute
        stream on the hop, fix-up pass (GPU microseconds from events on the compute stream), and the host seconds
        spent waiting on the hop."""
        if not self._marks:
            return None
        torch.cuda.synchronize()
        scan = hop = fix = host = tr = 0.0
        n = 0
        for m in self._marks:
            if m["scan"][0] is None:
                continue
            scan += m["scan"][0].elapsed_time(m["scan"][1]) * 1e3
            hop += m["hop"][0].elapsed_time(m["hop"][1]) * 1e3 if m["hop"][0] is not None else 0.0
            fix += m["fix"][0].elapsed_time(m["fix"][1]) * 1e3 if m["fix"][0] is not None else 0.0
            host += m["hop_host_s"]
            tr += m.get("transport_s") or 0.0
            n += 1
        if not n:
            return None
        tot = scan + hop + fix
        return {"rounds": n, "chunk": self.shard.chunk, "overlap": self.overlap, "parallel_hop": self.parallel_hop,
                "scan_us": round(scan / n, 1), "hop_stall_us": round(hop / n, 1), "fixup_us": round(fix / n, 1),
                "hop_host_wait_us": round(host / n * 1e6, 1),
                "hop_transport_us": round(tr / n * 1e6, 1) if tr else None,      # staged rehearsal only: gloo's host-side transfer of the frame
                "hop_plus_fixup_share": round((hop + fix) / tot, 4) if tot > 0 else None,
                "fixup_frames": min(self.shard.chunk, settle_frames(self.p, 2.0 ** -26))}

    # ---- overlapped schedule ---------------------------------------------------------------------------------------
    def submit_round(self, frames: Optional[torch.Tensor], round_index: int, active: Optional[int] = None):
        """Overlapped form of run_round: returns the list of (round_index, out) that became final in this call
        (the previous round's, or this round's own when it needs no hop).  Falls back to run_round when the schedule
        cannot overlap (world 1, p = 0, exact ring chain)."""
        if not self.overlap:
            out = self.run_round(frames, round_index, active)
            return [(round_index, out)] if out is not None else []
        sh, p = self.shard, self.p
        w, r = sh.world, sh.rank
        a = w if active is None else int(active)
        if not (1 <= a <= w):
            raise ValueError(f"active = {a} outside 1..{w}")
        has_frames = frames is not None and frames.shape[0] > 0 and r < a
        done = []
        if not has_frames:
            done += self.flush()
            return done
        c = sh.chunk_of(round_index)
        first = c * sh.chunk
        slot = round_index % max(1, int(getattr(self.engine, "slots", 1)))      # 2 slots: rounds alternate; 3 (the CLI): one more round may still be downloading
        e0 = self._ev(frames)
        if getattr(self.engine, "slots", 1) >= 2:
            local, out = self.engine.local_scan(frames, first, clip_start=(c == 0), slot=slot)
        else:
            local, out = self.engine.local_scan(frames, first, clip_start=(c == 0))
        e1 = self._ev(frames)
        n = frames.shape[0]
        final_local = local[n - 1]
        full = a == w
        dst = (r + 1) % w if (full or r + 1 < a) else None
        src = (r - 1) % w if (full or r > 0) else None
        rec = {"round": round_index, "c": c, "n": n, "local": local, "out": out, "full": full, "final": final_local,
               "src": src, "dst": dst, "scan": (e0, e1), "posted": None}
        if self.loopback or not self._staged(final_local):
            # RCCL: the transfer is enqueued behind the scan now and runs beside whatever the compute stream does next
            rec["posted"] = self._post(final_local if dst is not None else None, final_local, src, dst)
        else:                               # gloo + device tensors: staged through pinned host memory on a worker thread
            rec["posted"] = self._post_staged_async(final_local if dst is not None else None, final_local, src, dst)
        prev, self._pending = self._pending, rec
        if prev is not None:
            done.append(self._finish(prev))
        return done

    def _finish(self, rec):
        p, r = self.p, self.shard.rank
        final_local = rec["final"]
        t0 = None
        h0 = self._ev(final_local)
        import time as _time
        t0 = _time.perf_counter()
        got = self._complete(*rec["posted"], final_local.device)
        host_s = _time.perf_counter() - t0
        h1 = self._ev(final_local)
        if r == 0:
            carry = self.carry_next_round
            if rec["full"]:
                self.carry_next_round = got      # what arrived now seeds the next round
        else:
            carry = got
        if rec["c"] == 0:
            carry = None
        f0 = f1 = None
        if carry is not None:
            k = min(rec["n"], settle_frames(p, 2.0 ** -26))
            f0 = self._ev(final_local)
            self.engine.correct(rec["local"][:k], carry, p, rec["out"][:k])
            f1 = self._ev(final_local)
        if self.timing:
            self._mark({"scan": rec["scan"], "hop": (h0, h1), "fix": (f0, f1), "hop_host_s": host_s,
                        "transport_s": self._last_transport_s})
        return rec["round"], rec["out"]

    def flush(self):
        """Finish the round still in flight (overlapped mode); [] otherwise."""
        if self._pending is None:
            return []
        rec, self._pending = self._pending, None
        return [self._finish(rec)]

    def run_round(self, frames: Optional[torch.Tensor], round_index: int, active: Optional[int] = None):
        """frames: this rank's chunk for this round (None when it owns none).  `active` = how many ranks own a chunk in
        this round (default: all; fewer only in the last round of a clip that is not a multiple of world * chunk frames —
        every rank must pass the same value, see `active_ranks`).  The last chunk of the clip may be shorter than
        `chunk`.  Returns the finished uint8 frames (None for a rank without a chunk)."""
        sh, p = self.shard, self.p
        w, r = sh.world, sh.rank
        a = w if active is None else int(active)
        if not (1 <= a <= w):
            raise ValueError(f"active = {a} outside 1..{w}")
        if self.overlap:
            raise RuntimeError("overlap=True: use submit_round()/flush() (results arrive one call late)")
        has_frames = frames is not None and frames.shape[0] > 0 and r < a
        c = sh.chunk_of(round_index)
        first = c * sh.chunk
        # an engine with several output slots (GpuShardEngine(slots=3) of the CLI, whose download stream may still be reading round r - 1's
        # frames) gets a fresh slot every round in THIS schedule too — the default over RCCL, and what every p = 0 render runs
        nslots = int(getattr(self.engine, "slots", 1))
        slot_kw = {"slot": round_index % nslots} if nslots >= 2 else {}
        if w == 1 and not self.loopback and p > 0.0 and hasattr(self.engine, "sequential_scan"):
            # one rank owns consecutive chunks: carry the state itself (the reference's in-order loop, ref:1081-1105),
            # no zero-state scan and no correction pass
            out, self.carry_next_round = self.engine.sequential_scan(frames, first, None if c == 0 else self.carry_next_round, **slot_kw)
            return out
        if not has_frames:
            # a partial last round: nothing to render here, and nobody downstream waits for this rank's state
            # (the exchange below is only between ranks < active, and there is no next round to seed)
            return None
        e0 = self._ev(frames)
        local, out = self.engine.local_scan(frames, first, clip_start=(c == 0), **slot_kw)
        e1 = self._ev(frames)
        if p <= 0.0:
            return out
        n = frames.shape[0]
        final_local = local[n - 1]
        full = a == w                       # a full round also seeds rank 0 for the next round
        carry = None
        import time as _time
        h0 = self._ev(final_local)
        t0 = _time.perf_counter()
        if w == 1 and not self.loopback:
            carry = self.carry_next_round
            true_final = final_local if carry is None else final_local + (p ** n) * carry
            self.carry_next_round = true_final.clone()
        elif self.parallel_hop:
            # every rank forwards its chunk-final local state (== true state to float32 rounding; only full-length
            # chunks ever send: a short chunk is the last of the clip)
            dst = (r + 1) % w if (full or r + 1 < a) else None
            src = (r - 1) % w if (full or r > 0) else None
            got = self._send_recv(final_local if dst is not None else None, final_local, src, dst)
            if r == 0:
                carry = self.carry_next_round
                if full:
                    self.carry_next_round = got      # what arrived now seeds the next round
            else:
                carry = got
            if c == 0:
                carry = None
        else:
            # exact chain: true finals travel down the ring
            if r == 0:
                carry = None if c == 0 else self.carry_next_round
                true_final = final_local if carry is None else final_local + (p ** n) * carry
                if w == 1:                  # loopback: the ring's one hop, rank 0 to itself (send and receive in one group)
                    self.carry_next_round = self._send_recv(true_final, final_local, 0, 0)
                else:
                    if a > 1:
                        self._send_recv(true_final, final_local, None, 1)
                    if full:
                        self.carry_next_round = self._send_recv(None, final_local, w - 1, None)
            else:
                carry = self._send_recv(None, final_local, r - 1, None)
                true_final = final_local + (p ** n) * carry
                if full or r + 1 < a:
                    self._send_recv(true_final, final_local, None, (r + 1) % w)
        host_s = _time.perf_counter() - t0
        h1 = self._ev(final_local)
        f0 = f1 = None
        if carry is not None:
            # p^(j+1) * carry is below float32 resolution of the state scale after settle_frames(p) frames: the frames
            # behind that point keep the bytes of the local scan (a 1-LSB flip there needs the local value within
            # 2^-26 * 255 of a rounding boundary)
            k = min(n, settle_frames(p, 2.0 ** -26))
            f0 = self._ev(final_local)
            self.engine.correct(local[:k], carry, p, out[:k])
            f1 = self._ev(final_local)
        if self.timing:
            self._mark({"scan": (e0, e1), "hop": (h0, h1), "fix": (f0, f1), "hop_host_s": host_s})
        return out
