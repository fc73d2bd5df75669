"""The hot slice of process_video (crt_filter.py ref:919-920, :1037-1131) over device-resident
frames: masks built once per render, phase = i/fps*speed (ref:1043), time_sec = i/fps (ref:1064),
frame-parallel static effects, in-order persistence IIR (ref:1086-1096) and uint8 quantise
(ref:1098).  The reference parallelises frames over <= 2 worker threads (ref:1015-1017); here a
run of frames is enqueued back to back on one GPU, and runs are sharded across GPUs by
`FrameShard` (one process per GPU; RCCL only for the one-frame persistence carry).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib, tables
from .effects import Engine, Settings, _overlay_tensor, make_triad_mask, make_vignette


_FRAME_DTYPE = np.dtype(_lib.CrtfxFrame)      # crtfx_frame as a numpy record (same layout: numpy takes it from the ctypes struct)


_GLITCH_POOL = None


def _glitch_pool():
    """A small thread pool for the per-frame numpy RNG draws of the glitch band (host work of the render loop)."""
    global _GLITCH_POOL
    if _GLITCH_POOL is None:
        import concurrent.futures
        import os
        _GLITCH_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)), thread_name_prefix="crtfx-glitch")
    return _GLITCH_POOL


@dataclass
class RenderSettings:
    """process_video's effect keywords (ref:864-911) with the CLI defaults (ref:1155-1206)."""
    scanline_strength: float = 0.6
    triad_strength: float = 0.35
    triad_gamma: float = 2.2
    triad_preserve_luma: bool = False
    triad_softness: float = 0.5
    aberration_px: int = 1
    bloom_sigma: float = 1.2
    bloom_strength: float = 0.25
    bloom_threshold: float = 0.0
    noise_strength: float = 1.5
    vignette_strength: float = 0.25
    persistence: float = 0.2
    scanline_speed_px_s: float = 30.0
    scanline_period_px: float = 2.0
    fast_bloom: bool = True
    pixel_size: int = 2
    brightness: float = 0.0
    contrast: float = 1.0
    gamma: float = 1.0
    saturation: float = 1.0
    temperature: float = 0.0
    flicker_strength: float = 0.0
    flicker_hz: float = 0.0
    grain_size: int = 1
    scanline_angle: float = 0.0
    scanline_thickness: float = 1.0
    warp_strength: float = 0.0
    glitch_amp_px: int = 0
    glitch_height_frac: float = 0.0

    def static_settings(self, h: int, w: int) -> Settings:
        """ref:919-920 — masks are built once per render, None when the strength is 0."""
        tm = make_triad_mask(h, w, self.triad_strength, self.triad_softness) if self.triad_strength > 0.0 else None
        vg = make_vignette(h, w, self.vignette_strength) if self.vignette_strength > 0.0 else None
        return Settings(
            scanline_strength=self.scanline_strength, triad_mask=tm, triad_gamma=float(self.triad_gamma),
            triad_preserve_luma=bool(self.triad_preserve_luma), aberration_px=self.aberration_px,
            bloom_sigma=self.bloom_sigma, bloom_strength=self.bloom_strength, bloom_threshold=float(self.bloom_threshold),
            noise_strength=self.noise_strength, vignette_mask=vg, scanline_period_px=self.scanline_period_px,
            fast_bloom=self.fast_bloom, pixel_size=self.pixel_size, brightness=float(self.brightness),
            contrast=float(self.contrast), gamma=float(self.gamma), saturation=float(self.saturation),
            temperature=float(self.temperature), flicker_strength=float(self.flicker_strength),
            flicker_hz=float(self.flicker_hz), grain_size=int(self.grain_size), scanline_angle=float(self.scanline_angle),
            scanline_thickness=float(self.scanline_thickness), warp_strength=float(self.warp_strength))


# BASELINE.json configs as RenderSettings
def baseline_config(n: int) -> Tuple[RenderSettings, int, int]:
    off = dict(triad_strength=0.0, aberration_px=0, bloom_strength=0.0, noise_strength=0.0, vignette_strength=0.0,
               persistence=0.0, fast_bloom=False, pixel_size=1)
    full = dict(scanline_strength=0.6, triad_strength=0.35, triad_gamma=2.2, triad_softness=0.5, triad_preserve_luma=False,
                aberration_px=1, bloom_strength=0.25, fast_bloom=False, warp_strength=0.15, vignette_strength=0.25,
                noise_strength=1.5, grain_size=1, pixel_size=1, persistence=0.0)
    if n == 1:
        return RenderSettings(**dict(off, scanline_strength=0.6, scanline_period_px=2.0, scanline_speed_px_s=30.0)), 720, 1280
    if n == 2:
        return RenderSettings(**dict(full, bloom_sigma=1.2)), 1080, 1920
    if n == 3:
        return RenderSettings(**dict(full, bloom_sigma=3.0)), 2160, 3840
    if n == 4:
        return RenderSettings(**dict(full, bloom_sigma=1.2, persistence=0.5)), 1080, 1920
    if n == 0:      # not a BASELINE config: the reference CLI's default flag set (ref:1160-1206) at 1080p
        return RenderSettings(), 1080, 1920
    if n == 5:      # 8K, as config 3, pixels held as fp16 in and out
        return RenderSettings(**dict(full, bloom_sigma=3.0)), 4320, 7680
    raise ValueError(f"unknown BASELINE config {n}")


class FramePipeline:
    """One GPU's worth of the render loop."""

    def __init__(self, device: torch.device, h: int, w: int, settings: RenderSettings, fps: float = 30.0,
                 noise_seed: int = 0, dtype: torch.dtype = torch.uint8, text_overlay_rgba=None, text_overlay_after: bool = True):
        self.device, self.h, self.w, self.rs, self.fps = device, int(h), int(w), settings, float(fps)
        # the one overlay plane of a render (ref:1076-1077: built once per frame there, identical every time)
        self.overlay = _overlay_tensor(text_overlay_rgba, device, int(h), int(w)) if text_overlay_rgba is not None else None
        self.overlay_after = bool(text_overlay_after)
        self.noise_seed = int(noise_seed)
        self.dtype = dtype                  # torch.uint8, or torch.float16 (half frames on the 0..255 scale)
        self._glitch_stage = {}             # (frames, rows, segments) -> {"slots": two [pinned int32 block, upload event] pairs, "turn": which one is next} (frame_records)
        self.engine = Engine(device, h, w, _lib.PIX_F16 if dtype == torch.float16 else _lib.PIX_U8)
        self.static = settings.static_settings(self.h, self.w)
        self.engine.set_params(self.static)
        self.lib = self.engine.lib
        self._scan_cache = {}

    # ---- per-frame records for frames [first, first+n) -------------------------------------
    def frame_records(self, first: int, n: int, noise_planes: Optional[torch.Tensor] = None):
        """-> (records, keep-alive list).  The records are a numpy array laid out as crtfx_frame[n], filled column by
        column (no per-frame Python work on the common path: at 4K the GPU finishes a frame in ~0.1 ms)."""
        rs, st = self.rs, self.static
        idx = np.arange(first, first + n, dtype=np.int64)
        recs = np.zeros(n, dtype=_FRAME_DTYPE)
        hold = [recs]
        t_sec = idx.astype(np.float64) / float(self.fps)                                            # ref:1064, i / fps
        if st.scanline_strength > 0.0:
            phases = t_sec * rs.scanline_speed_px_s                                                 # ref:1043
            if st.scanline_angle == 0.0 and st.scanline_thickness == 1.0:
                table, offs = self._scan_rows(phases)
                hold.append(table)                      # the records point into it: it lives as long as they do
                recs["scan_row_dev"] = table.data_ptr() + offs * 4
            else:
                # slanted / shaped scanlines: the reference rebuilds an H x W float64 sin/pow mask per frame on the CPU
                # (ref:308-328); here one small kernel per frame writes it on the device (crtfx_scanline_plane)
                planes = torch.empty((n, self.h, self.w), dtype=torch.float32, device=self.device)
                omega, tan_t, inv_sharp = tables.scanline_plane_scalars(st.scanline_period_px, st.scanline_angle, st.scanline_thickness)
                stream = torch.cuda.current_stream(self.device).cuda_stream
                with torch.cuda.device(self.device):
                    for j, ph in enumerate(phases):
                        _lib.check(self.lib, self.engine.ctx, self.lib.crtfx_scanline_plane(
                            self.engine.ctx, float(st.scanline_strength), omega, float(ph), tan_t, inv_sharp, planes[j].data_ptr(), stream))
                hold.append(planes)
                recs["scan_plane_dev"] = planes.data_ptr() + np.arange(n, dtype=np.uint64) * np.uint64(self.h * self.w * 4)
        flick = (self.engine.flags & _lib.F_FLICKER) != 0
        if rs.glitch_amp_px > 0 and rs.glitch_height_frac > 0.0:                                    # ref:835-859, render variant
            # every frame draws its band's offsets from its own numpy Generator (ref:842-849): ~0.3 ms of host time per 1080p frame
            # — drawn on a few threads (the Generator releases the GIL), gathered in ONE pinned buffer, uploaded once per batch
            def draw(i):
                ph = (int(i) / float(self.fps)) * rs.scanline_speed_px_s
                return tables.glitch_offsets_render_segments(self.h, self.w, ph, rs.glitch_amp_px, rs.glitch_height_frac)
            drawn = list(_glitch_pool().map(draw, idx)) if n > 1 else [draw(idx[0])]
            if drawn and drawn[0][1] is not None:
                rows, cols = drawn[0][1].shape
                # two pinned staging blocks per shape, reused in turn (a fresh hipHostMalloc per batch made one step in three take 15-80 ms
                # instead of 6: profiles/r04_cli_scan.txt); a block is rewritten only once the upload that last read it has completed
                shape = self._glitch_stage.setdefault((n, rows, cols), {"slots": [], "turn": 0})      # the turn is per shape: two shapes that
                slots = shape["slots"]                                                                 # alternate (a full batch and a short last one) must not pin each other to one slot
                if len(slots) < 2:
                    slots.append([torch.empty((n, rows, cols), dtype=torch.int32).pin_memory(), None])
                    host, ev = slots[-1][0], None
                    k = len(slots) - 1
                else:
                    k = shape["turn"] = (shape["turn"] + 1) & 1
                    host, ev = slots[k]
                if ev is not None:
                    ev.synchronize()
                hv = host.numpy()
                for j, (_, offs, _) in enumerate(drawn):
                    hv[j] = offs
                dev = host.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                slots[k][1] = ev
                hold += [dev]
                recs["glitch_offs_dev"] = dev.data_ptr() + np.arange(n, dtype=np.uint64) * np.uint64(rows * cols * 4)
                recs["glitch_y0"], recs["glitch_cols"], recs["glitch_seg_len"] = int(drawn[0][0]), int(cols), int(drawn[0][2])
        if flick:
            recs["flicker_factor"] = [tables.flicker_factor(st.flicker_strength, st.flicker_hz, int(i) / float(self.fps)) for i in idx]   # ref:1064
        else:
            recs["flicker_factor"] = 1.0
        recs["noise_seed"] = np.uint64(self.noise_seed & 0xFFFFFFFFFFFFFFFF)
        recs["frame_index"] = idx.astype(np.uint64)
        if self.overlay is not None:
            recs["overlay_rgba_dev"], recs["overlay_after"] = self.overlay.data_ptr(), int(self.overlay_after)
        if noise_planes is not None:
            recs["noise_plane_dev"] = [noise_planes[j].data_ptr() for j in range(n)]
            hold.append(noise_planes)
        return recs, hold

    def _scan_rows(self, phases: np.ndarray):
        """The scanline row gains of every frame of a batch: (device table, float offset per frame into it).
        The reference evaluates sin on float32(y) + float32(phase) (ref:213-217).  When every phase of the batch is an
        integer (scanline_speed a multiple of fps: the CLI default 30 / 30) those sums are the integers y + phase
        exactly, so ONE table g[k], k = min phase .. max phase + H - 1, holds every row of every frame — frame b's rows
        are g[phase_b - min phase : ... + H], the same float32 values np.sin gives the reference — and a batch costs
        H + n sines instead of H * n.  Otherwise: one (n, H) table per batch, as before."""
        st = self.static
        ph32 = phases.astype(np.float32)
        if np.all(ph32 == np.rint(ph32)) and float(np.abs(ph32).max(initial=0.0)) + self.h + 70000 < 2.0 ** 24:
            lo = int(ph32.min())
            hi = int(ph32.max())
            hit = self._scan_cache.get("table")
            if hit is None or lo < hit[0] or hi + self.h > hit[1]:
                # a table that also covers the batches to come (a render's phases only grow): a host -> device copy from
                # pageable memory waits for everything already queued on the stream, i.e. it would serialise the host
                # with the GPU once per batch
                top = hi + self.h + 65536
                k = np.arange(lo, top, dtype=np.float32)
                g = tables.scanline_rows_at(k, st.scanline_strength, st.scanline_period_px)      # scanline_rows' expression on the sums themselves
                hit = (lo, top, torch.from_numpy(g).pin_memory().to(self.device, non_blocking=True))
                self._scan_cache["table"] = hit
            return hit[2], (ph32.astype(np.int64) - hit[0]).astype(np.uint64)
        rows = torch.from_numpy(tables.scanline_rows(self.h, st.scanline_strength, st.scanline_period_px, phases)).to(self.device)
        return rows, np.arange(len(phases), dtype=np.uint64) * np.uint64(self.h)

    def run(self, frames: torch.Tensor, first_index: int = 0, state: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None, noise_planes: Optional[torch.Tensor] = None,
            records=None, local_states: Optional[torch.Tensor] = None, force_blend_first: bool = False):
        """frames: uint8 (N, H, W, 3) on self.device.  Returns (out uint8 (N, H, W, 3), state).
        `state` is the persistence carry (float32 H x W x 3) from the previous run, or None at the
        start of the clip (the first frame then passes through unblended, ref:1094-1095)."""
        n = frames.shape[0]
        assert frames.dtype == self.dtype and tuple(frames.shape[1:]) == (self.h, self.w, 3) and frames.is_contiguous()
        if out is None:
            out = torch.empty_like(frames)
        p = float(self.rs.persistence)
        recs, hold = records if records is not None else self.frame_records(first_index, n, noise_planes)
        has_state = state is not None
        if p > 0.0 and state is None:
            state = torch.empty((self.h, self.w, 3), dtype=torch.float32, device=self.device)
        stride = self.h * self.w * 3 * frames.element_size()
        if isinstance(recs, np.ndarray):
            recs = recs.ctypes.data_as(ctypes.POINTER(_lib.CrtfxFrame))
        with torch.cuda.device(self.device):
            rc = self.lib.crtfx_process_batch(
                self.engine.ctx, frames.data_ptr(), stride, out.data_ptr(), stride, n, recs,
                state.data_ptr() if (p > 0.0) else None, p, 1 if (has_state or force_blend_first) else 0,
                local_states.data_ptr() if local_states is not None else None,
                torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib, self.engine.ctx, rc)
        self._hold = hold                         # keep per-frame tables (scanline gains included) alive until the next run replaces them
        return out, (state if p > 0.0 else None)

    def plan(self) -> dict:
        """Which kernel builds the last run() landed on (crtfx_last_plan): {"phosphor": "k_phosphor_ct<9,u8>", "group": 2, "seg_rows": 256,
        "warp": "k_warp_lean<...,plain>", ...}."""
        return self.engine.last_plan()

    # ---- profiling hooks (HIP events recorded by the library on the launch stream) ----------
    def profile(self, on):
        """on: False/0 off, True/1 time every launch, N > 1 time the launches of every N-th frame."""
        _lib.check(self.lib, self.engine.ctx, self.lib.crtfx_profile_enable(self.engine.ctx, int(on)))

    def profile_read(self):
        out = {}
        for k, name in ((0, "k_phosphor"), (1, "k_warp"), (2, "k_bloom_pass")):
            ms, cnt, fr = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
            _lib.check(self.lib, self.engine.ctx,
                       self.lib.crtfx_profile_read(self.engine.ctx, k, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fr)))
            out[name] = (ms.value, cnt.value, fr.value)      # mean ms per launch, timed launches, frames they covered
        return out


# ---------------------------------------------------------------------------------------
# GPU engine for shard.ShardedRender: local scan = crtfx_process_batch from a zero
# state, correction = crtfx_halo_correct_quantise per frame.
# ---------------------------------------------------------------------------------------

class _LocalStates:
    """What ShardedRender reads of a chunk's local states: the first k of them (the frames the fix-up re-quantises) and the
    chunk-final one (the frame that travels to the next rank) — indexable like the (n, H, W, 3) tensor they stand for."""

    def __init__(self, first: torch.Tensor, final: torch.Tensor, n: int):
        self.first, self.final, self.n = first, final, int(n)
        self.shape = (self.n,) + tuple(final.shape)

    def __getitem__(self, key):
        if isinstance(key, slice):
            stop = self.n if key.stop is None else key.stop
            if (key.start or 0) != 0 or key.step not in (None, 1) or stop > self.first.shape[0]:
                raise IndexError(f"local states kept for frames [0, {self.first.shape[0]}) and frame {self.n - 1} only (asked: {key})")
            return self.first[:stop]
        k = int(key)
        if k in (self.n - 1, -1):
            return self.final
        return self.first[k]


class GpuShardEngine:
    """Buffers of one rank's chunks.  `slots` = 2 double-buffers the per-frame local states and the output frames so that
    round r+1's scan can be enqueued while round r's state frame is still travelling (ShardedRender overlap=True).
    Per-frame float32 local states are kept for the frames the fix-up can change only — the first settle_frames(p, 2^-26)
    of a chunk (26 at p = 0.5, the whole chunk when it is shorter than that: the exact-chain schedule) — plus the
    chunk-final state; the rest of the chunk runs with its state in registers (crtfx_process_batch's runs)."""

    def __init__(self, pipe: FramePipeline, chunk: int, slots: int = 1):
        from .shard import settle_frames
        self.pipe = pipe
        self.slots = int(slots)
        h, w = pipe.h, pipe.w
        p = pipe.rs.persistence
        self.keep = min(int(chunk), settle_frames(p, 2.0 ** -26)) if p > 0.0 else 0
        self.local = [torch.empty((self.keep, h, w, 3), dtype=torch.float32, device=pipe.device) for _ in range(self.slots)] if p > 0.0 else None
        self.final = [torch.empty((h, w, 3), dtype=torch.float32, device=pipe.device) for _ in range(self.slots)] if p > 0.0 else None
        self.out_slots = [torch.empty((chunk, h, w, 3), dtype=pipe.dtype, device=pipe.device) for _ in range(self.slots)]
        self.out = self.out_slots[0]
        self.zero = torch.zeros((h, w, 3), dtype=torch.float32, device=pipe.device) if p > 0.0 else None
        self.state = torch.empty((h, w, 3), dtype=torch.float32, device=pipe.device) if p > 0.0 else None
        self.records = {}

    def local_scan(self, frames, first_index, clip_start, slot: int = 0):
        n = frames.shape[0]
        recs = self.records.pop(first_index, None)
        out = self.out_slots[slot % self.slots]
        if self.pipe.rs.persistence <= 0.0:
            self.pipe.run(frames, first_index=first_index, out=out[:n], records=recs)
            return None, out[:n]
        local = self.local[slot % self.slots]
        final = self.final[slot % self.slots]
        state = None
        if not clip_start:                                      # zero incoming state, blend from the first frame on
            state = self.state
            state.copy_(self.zero)
        if recs is None:
            recs = self.pipe.frame_records(first_index, n)
        arr, hold = recs
        k = min(n, self.keep)
        _, st = self.pipe.run(frames[:k], first_index=first_index, state=state, out=out[:k], records=(arr[:k], hold), local_states=local[:k])
        if n > k:                                               # the rest of the chunk continues from state k - 1 (left in st), no per-frame states
            _, st = self.pipe.run(frames[k:], first_index=first_index + k, state=st, out=out[k:n], records=(arr[k:], hold))
        final.copy_(st)                                         # its own buffer per slot: the next round's scan reuses self.state while this one travels
        return _LocalStates(local[:k], final, n), out[:n]

    def sequential_scan(self, frames, first_index, state, slot: int = 0):
        """world 1: the chunk continues from the true state of the previous one (None at the start of the clip)."""
        n = frames.shape[0]
        recs = self.records.pop(first_index, None)
        out, state = self.pipe.run(frames, first_index=first_index, state=state, out=self.out_slots[slot % self.slots][:n], records=recs)
        return out, state

    def correct(self, local, carry, p, out):
        """out[j] = quantise(clip(local[j] + p^(j+1) * carry)) for the whole chunk in one launch per 64 frames."""
        pipe = self.pipe
        n = int(local.shape[0])
        assert local.is_contiguous() and out.is_contiguous() and carry.is_contiguous()
        with torch.cuda.device(pipe.device):
            rc = pipe.lib.crtfx_halo_correct_batch(pipe.engine.ctx, local.data_ptr(), carry.data_ptr(), float(p), 1, n, out.data_ptr(),
                                                   out.stride(0) * out.element_size(), torch.cuda.current_stream(pipe.device).cuda_stream)
        _lib.check(pipe.lib, pipe.engine.ctx, rc)
