"""`process_frames` — the hot loop of the reference's `process_video` (crt_filter.py ref:1037-1131) behind a callable a maintainer can drop
into `process_video` itself: the frame iterator the reference already has (`clip.iter_frames(...)` ref:1036 or `FFmpegRawReader.iter_frames()`
ref:1034), the writer call it already makes (`writer.write_frame`, ref:1101) and the effect keywords `process_video` was called with
(ref:864-911, same names).  In `process_video` the lines ref:1015-1131 — thread pool, futures dictionary, in-order drain, persistence blend,
`convertScaleAbs` — become

    n = pythoncrt_amd.process_frames(frame_iter, writer.write_frame, out_w, out_h, fps_out, total_frames,
                                     scanline_strength=scanline_strength, triad_strength=triad_strength, ..., progress_cb=progress_cb)

What it keeps of the reference's loop: frames of another size are resized with Pillow's BILINEAR first (ref:1039-1041); frame i runs at
phase = i / fps * scanline_speed_px_s and time_sec = i / fps (ref:1043, :1064); frames are committed strictly in order, frame 0 passes through
unblended, frame i > 0 blends with the state frame i - 1 left (ref:1086-1096); `progress_cb(min(1, frames_written / total_frames))` after
every frame (ref:1104-1105); the text overlay is rasterised once (ref:1076-1077 builds the same plane for every frame).  What differs: the
frames of a batch go to the GPU together (pinned staging, upload / kernels / download on three streams, batch k's kernels under the host's
writes of batch k - 1 and reads of batch k + 1) instead of one `apply_static_effects` call per frame on two worker threads."""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Tuple

import numpy as np

# process_video keywords that belong to its container / codec plumbing (out of scope): accepted so that a caller can forward its
# own keyword dictionary unchanged, and ignored
_IO_KEYS = ("input_path", "output_path", "width", "height", "fps", "crf", "target_bitrate_kbps", "gpu", "nvenc_preset", "encoder_preference",
            "decoder_preference")


def iter_rgb24(stream, out_w: int, out_h: int):
    """The frame iterator of the reference's FFmpegRawReader.iter_frames (ref:495-506) over an ALREADY OPEN byte stream of raw rgb24 — the
    stdout of an ffmpeg process the caller started (`-f rawvideo -pix_fmt rgb24 -`), a file, a pipe: frames of out_h x out_w x 3 uint8 until the
    stream ends; a trailing partial frame is dropped, as there.  (The reader class itself — spawning ffmpeg, hw-accel flags — is codec plumbing
    and stays the reference's.)"""
    frame_size = int(out_w) * int(out_h) * 3
    while True:
        buf = stream.read(frame_size)
        while buf and len(buf) < frame_size:           # a pipe may return less than asked for: keep reading until the frame is whole or the stream ends
            more = stream.read(frame_size - len(buf))
            if not more:
                break
            buf += more
        if not buf or len(buf) < frame_size:
            return
        yield np.frombuffer(buf, dtype=np.uint8).reshape((int(out_h), int(out_w), 3))


def process_frames(frame_iter: Iterable[np.ndarray], write_frame: Callable[[np.ndarray], None], out_w: int, out_h: int, fps_out: float,
                   total_frames: Optional[int] = None, *,
                   scanline_strength: float = 0.6, triad_strength: float = 0.35, triad_gamma: float = 2.2, triad_preserve_luma: bool = False,
                   triad_softness: float = 0.5, aberration_px: int = 1, bloom_sigma: float = 1.2, bloom_strength: float = 0.25,
                   noise_strength: float = 1.5, vignette_strength: float = 0.25, persistence: float = 0.2, scanline_speed_px_s: float = 30.0,
                   scanline_period_px: float = 2.0, fast_bloom: bool = True, pixel_size: int = 2, glitch_amp_px: int = 0,
                   glitch_height_frac: float = 0.0, bloom_threshold: float = 0.0, brightness: float = 0.0, contrast: float = 1.0,
                   gamma: float = 1.0, saturation: float = 1.0, temperature: float = 0.0, flicker_strength: float = 0.0, flicker_hz: float = 0.0,
                   grain_size: int = 1, scanline_angle: float = 0.0, scanline_thickness: float = 1.0, warp_strength: float = 0.0,
                   text: str = "", text_font: str = "", text_size: int = 36, text_color: str = "#FFFFFF", text_pos: Tuple[int, int] = (32, 32),
                   text_after: bool = True, progress_cb: Optional[Callable[[float], None]] = None,
                   batch: int = 16, noise_seed: Optional[int] = None, device=None, **io_keywords) -> int:
    """Render every frame of `frame_iter` (H x W x 3 uint8 RGB arrays) and hand the finished uint8 frames to `write_frame` in order.
    Effect keywords: the names, meaning and defaults of process_video / the CLI (ref:864-911, :1155-1206); the caller applies the clamps of
    ref:1225-1266 as the reference's `main` does (`pythoncrt_amd.cli.settings_from_args` restates them).  Returns the number of frames written.
    The array passed to `write_frame` is a view of a staging buffer that is reused two batches later: consume it inside the call (the
    reference's `FFMPEG_VideoWriter.write_frame` writes it to the encoder's pipe at once).
    `progress_cb(fraction)` is called after every written frame with min(1, written / total_frames) (ref:1104-1105; the reference always knows
    its total, ref:1029).  With `total_frames=None` no fraction can be formed: the callback is then called ONCE, with 1.0, after the last frame.
    If `frame_iter` or `write_frame` raises, the GPU work already queued is drained (device synchronize) before the exception leaves this
    function, so that the staging buffers are not freed under a running copy."""
    import os
    import torch
    from .pipeline import FramePipeline, RenderSettings
    unknown = set(io_keywords) - set(_IO_KEYS)
    if unknown:
        raise TypeError(f"process_frames() got unexpected keyword arguments {sorted(unknown)}")
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm device visible; pythoncrt_amd has no CPU fallback")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    h, w, B = int(out_h), int(out_w), max(1, int(batch))
    rs = RenderSettings(
        scanline_strength=float(scanline_strength), triad_strength=float(triad_strength), triad_gamma=float(triad_gamma),
        triad_preserve_luma=bool(triad_preserve_luma), triad_softness=float(triad_softness), aberration_px=int(aberration_px),
        bloom_sigma=float(bloom_sigma), bloom_strength=float(bloom_strength), bloom_threshold=float(bloom_threshold),
        noise_strength=float(noise_strength), vignette_strength=float(vignette_strength), persistence=float(persistence),
        scanline_speed_px_s=float(scanline_speed_px_s), scanline_period_px=float(scanline_period_px), fast_bloom=bool(fast_bloom),
        pixel_size=int(pixel_size), brightness=float(brightness), contrast=float(contrast), gamma=float(gamma), saturation=float(saturation),
        temperature=float(temperature), flicker_strength=float(flicker_strength), flicker_hz=float(flicker_hz), grain_size=int(grain_size),
        scanline_angle=float(scanline_angle), scanline_thickness=float(scanline_thickness), warp_strength=float(warp_strength),
        glitch_amp_px=int(glitch_amp_px), glitch_height_frac=float(glitch_height_frac))
    overlay = None
    if text:                                                                   # ref:1076-1077 (the same plane for every frame: built once)
        from .text import make_text_overlay_rgba
        overlay = make_text_overlay_rgba(w, h, text, text_font, int(text_size), text_color, tuple(text_pos))
    seed = int(noise_seed) if noise_seed is not None else int.from_bytes(os.urandom(8), "little")
    pipe = FramePipeline(dev, h, w, rs, fps=float(fps_out), noise_seed=seed, text_overlay_rgba=overlay, text_overlay_after=bool(text_after))
    total = max(1, int(total_frames)) if total_frames else None

    NS = 2
    pin_in = [torch.empty((B, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(NS)]
    pin_out = [torch.empty((B, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(NS)]
    dev_in = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(NS)]
    dev_out = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(NS)]
    np_in = [t.numpy() for t in pin_in]
    np_out = [t.numpy() for t in pin_out]
    compute = torch.cuda.current_stream(dev)
    s_up, s_down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    up_done = [None] * NS            # the upload that last read pin_in[d]
    kernels_done = [None] * NS       # the kernels that last read dev_in[d] / wrote dev_out[d]
    down_done = [None] * NS          # the download that last read dev_out[d]
    state, index, written, k = None, 0, 0, 0
    pending = None                   # (slot, frames, download event) of the batch whose frames are still to be written

    def fit(frame):
        a = np.asarray(frame)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"frames must be H x W x 3 RGB arrays, got {a.shape}")
        if a.shape[0] != h or a.shape[1] != w:                                  # ref:1039-1041
            from PIL import Image
            a = np.asarray(Image.fromarray(np.ascontiguousarray(a, dtype=np.uint8)).resize((w, h), Image.BILINEAR))
        return a

    def drain(p):
        nonlocal written
        d, n, ev = p
        ev.synchronize()
        for j in range(n):
            write_frame(np_out[d][j])                                           # ref:1101
            written += 1
            if progress_cb is not None and total:
                progress_cb(min(1.0, written / float(total)))                   # ref:1104-1105

    try:
        it = iter(frame_iter)
        done = False
        while not done:
            d = k % NS
            if up_done[d] is not None:
                up_done[d].synchronize()             # the upload that read this pinned slot two batches ago (long done)
            n = 0
            while n < B:
                try:
                    frame = next(it)
                except StopIteration:
                    done = True
                    break
                np.copyto(np_in[d][n], fit(frame), casting="unsafe")
                n += 1
            if n:
                if kernels_done[d] is not None:
                    s_up.wait_event(kernels_done[d])
                with torch.cuda.stream(s_up):
                    dev_in[d][:n].copy_(pin_in[d][:n], non_blocking=True)
                    up = torch.cuda.Event()
                    up.record(s_up)
                up_done[d] = up
                compute.wait_event(up)
                if down_done[d] is not None:
                    compute.wait_event(down_done[d])
                _, state = pipe.run(dev_in[d][:n], first_index=index, state=state, out=dev_out[d][:n])
                kd = torch.cuda.Event()
                kd.record(compute)
                kernels_done[d] = kd
                # pin_out[d] was drained one iteration ago (drain below runs before the next batch is enqueued into the same slot)
                s_down.wait_event(kd)
                with torch.cuda.stream(s_down):
                    pin_out[d][:n].copy_(dev_out[d][:n], non_blocking=True)
                    dn = torch.cuda.Event()
                    dn.record(s_down)
                down_done[d] = dn
                index += n
            # the PREVIOUS batch's frames go to the writer while this batch is on the GPU
            if pending is not None:
                drain(pending)
                pending = None
            if n:
                pending = (d, n, dn)
            k += 1
        if pending is not None:
            drain(pending)
    except BaseException:
        try:
            torch.cuda.synchronize(dev)          # uploads / kernels / downloads still in flight reference the buffers above
        except Exception:       # noqa: BLE001 - the caller's exception is the one to report
            pass
        raise
    if progress_cb is not None and not total:
        progress_cb(1.0)
    return written
