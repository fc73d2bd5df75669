"""Command line of the reference (crt_filter.py parse_args ref:1153-1207, clamps in main
ref:1220-1267) over the GPU pipeline.

Same flag names, defaults and clamps: `build_parser` restates the flag schema of `parse_args` (ref:1154-1206) and `settings_from_args` the
clamp list of `main()` (ref:1225-1266) — the drop-in contract IS those names, defaults and bounds, so these two blocks follow
crt_filter.py (PythonCRT, GPL-3.0) line for line and are pinned against it by tests/golden/reference_cli.json.  What differs is the container I/O, which is out of scope
here (codec plumbing): frames are read and written as raw `rgb24` (H x W x 3 uint8,
the wire format of the reference's own FFmpegRawReader, ref:489-502, and of its ffmpeg writer
pipe), from a file or stdin/stdout, so the drop-in sits between two ffmpeg processes:

    ffmpeg -i in.mp4 -f rawvideo -pix_fmt rgb24 - |
      python -m pythoncrt_amd.cli --input - --width 1920 --height 1080 --fps 30 --output - [effect flags] |
      ffmpeg -f rawvideo -pix_fmt rgb24 -s 1920x1080 -r 30 -i - out.mp4

`--gui`, `--gpu`, `--nvenc-preset`, `--encoder`, `--decoder`, `--crf` and `--bitrate` are accepted for
compatibility and ignored (encode/decode/UI are not part of this path).  `--text*` rasterise the overlay on
the host with Pillow (ref:366-414) and alpha-blend it on the GPU before or after the effects.

Host staging: two pinned input and two pinned output batches; batch k's upload, kernels
and download are enqueued on one stream while the host writes batch k-1 and reads batch k+1.
"""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="pythoncrt_amd.cli", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--input", type=str, default="")
    p.add_argument("--output", type=str)
    p.add_argument("--width", type=int, default=0)
    p.add_argument("--height", type=int, default=0)
    p.add_argument("--fps", type=int, default=0)
    p.add_argument("--scanline-strength", type=float, default=0.6)
    p.add_argument("--triad-strength", type=float, default=0.35)
    p.add_argument("--triad-gamma", type=float, default=2.2)
    p.add_argument("--triad-preserve-luma", action="store_true")
    p.add_argument("--triad-softness", type=float, default=0.5)
    p.add_argument("--aberration-px", type=int, default=1)
    p.add_argument("--bloom-sigma", type=float, default=1.2)
    p.add_argument("--bloom-strength", type=float, default=0.25)
    p.add_argument("--bloom-threshold", type=float, default=0.0)
    p.add_argument("--noise-strength", type=float, default=1.5)
    p.add_argument("--vignette-strength", type=float, default=0.25)
    p.add_argument("--persistence", type=float, default=0.2)
    p.add_argument("--crf", type=int, default=18)
    p.add_argument("--bitrate", type=int, default=0)
    p.add_argument("--scanline-speed", type=float, default=30.0)
    p.add_argument("--scanline-period", type=float, default=2.0)
    p.add_argument("--fast-bloom", action="store_true")
    p.add_argument("--no-fast-bloom", dest="fast_bloom", action="store_false")
    p.set_defaults(fast_bloom=True)
    p.add_argument("--pixel-size", type=int, default=2)
    p.add_argument("--brightness", type=float, default=0.0)
    p.add_argument("--contrast", type=float, default=1.0)
    p.add_argument("--gamma", type=float, default=1.0)
    p.add_argument("--saturation", type=float, default=1.0)
    p.add_argument("--temperature", type=float, default=0.0)
    p.add_argument("--flicker-strength", type=float, default=0.0)
    p.add_argument("--flicker-hz", type=float, default=0.0)
    p.add_argument("--grain-size", type=int, default=1)
    p.add_argument("--scanline-angle", type=float, default=0.0)
    p.add_argument("--scanline-thickness", type=float, default=1.0)
    p.add_argument("--warp-strength", type=float, default=0.0)
    p.add_argument("--text", type=str, default="")
    p.add_argument("--text-font", type=str, default="")
    p.add_argument("--text-size", type=int, default=36)
    p.add_argument("--text-color", type=str, default="#FFFFFF")
    p.add_argument("--text-x", type=int, default=32)
    p.add_argument("--text-y", type=int, default=32)
    p.add_argument("--text-after", action="store_true")
    p.add_argument("--gpu", action="store_true")
    p.add_argument("--nvenc-preset", type=str, default="p4")
    p.add_argument("--encoder", type=str, default="auto", choices=["auto", "nvidia", "amd", "cpu"])
    p.add_argument("--decoder", type=str, default="auto", choices=["auto", "nvidia", "amd", "intel", "cpu"])
    p.add_argument("--glitch-amp", type=int, default=0)
    p.add_argument("--glitch-height", type=float, default=0.0)
    p.add_argument("--gui", action="store_true")
    # not in the reference
    p.add_argument("--batch", type=int, default=16, help="frames enqueued per GPU batch")
    p.add_argument("--noise-seed", type=int, default=None, help="seed of the counter-based grain RNG (default: random)")
    return p


def settings_from_args(a):
    """The clamps of main() (ref:1225-1266) -> RenderSettings."""
    from .pipeline import RenderSettings
    return RenderSettings(
        scanline_strength=float(max(0.0, min(1.0, a.scanline_strength))),
        triad_strength=float(max(0.0, min(1.0, a.triad_strength))),
        triad_gamma=float(max(0.1, a.triad_gamma)),
        triad_preserve_luma=bool(a.triad_preserve_luma),
        triad_softness=float(max(0.0, a.triad_softness)),
        aberration_px=int(max(-8, min(8, a.aberration_px))),
        bloom_sigma=max(0.0, a.bloom_sigma),
        bloom_strength=max(0.0, a.bloom_strength),
        noise_strength=max(0.0, a.noise_strength),
        vignette_strength=float(max(0.0, min(1.0, a.vignette_strength))),
        persistence=float(max(0.0, min(0.95, a.persistence))),
        scanline_speed_px_s=float(a.scanline_speed),
        scanline_period_px=max(1.0, float(a.scanline_period)),
        fast_bloom=bool(a.fast_bloom),
        pixel_size=max(1, int(a.pixel_size)),
        glitch_amp_px=max(0, int(a.glitch_amp)),
        glitch_height_frac=float(max(0.0, min(1.0, a.glitch_height))),
        bloom_threshold=float(max(0.0, min(1.0, a.bloom_threshold))),
        brightness=float(a.brightness),
        contrast=float(a.contrast),
        gamma=float(max(1e-3, a.gamma)),
        saturation=float(max(0.0, a.saturation)),
        temperature=float(max(-1.0, min(1.0, a.temperature))),
        flicker_strength=float(max(0.0, min(1.0, a.flicker_strength))),
        flicker_hz=float(max(0.0, a.flicker_hz)),
        grain_size=max(1, int(a.grain_size)),
        scanline_angle=float(a.scanline_angle),
        scanline_thickness=float(max(0.1, a.scanline_thickness)),
        warp_strength=float(max(-1.0, min(1.0, a.warp_strength))),
    )


def _read_into(fin, view: memoryview) -> int:
    """Fill `view` from a file or pipe; returns the bytes read (short only at end of stream)."""
    got = 0
    while got < len(view):
        k = fin.readinto(view[got:])
        if not k:
            break
        got += k
    return got


_IO_POOL = None
_IO_SLICE = 16 << 20            # bytes per positional read / write call


def _io_pool():
    """Threads for positional file I/O: one os.preadv / os.pwrite moves ~8 GB/s out of / into the page cache (a memcpy on one core,
    GIL released), a 4K frame is 24.9 MB each way — the raw rgb24 edge, not the GPU, would set the CLI's frame rate."""
    global _IO_POOL
    if _IO_POOL is None:
        import concurrent.futures
        import os
        _IO_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)), thread_name_prefix="crtfx-io")
    return _IO_POOL


def _pread_full(fd: int, view: memoryview, offset: int) -> int:
    """Fill `view` from `fd` at `offset` (regular file) in parallel slices; returns the bytes read (short only at end of file)."""
    import os

    def one(lo):
        hi, got = min(len(view), lo + _IO_SLICE), 0
        while lo + got < hi:
            k = os.preadv(fd, [view[lo + got:hi]], offset + lo + got)
            if k <= 0:
                break
            got += k
        return got
    starts = range(0, len(view), _IO_SLICE)
    parts = list(_io_pool().map(one, starts)) if len(view) > _IO_SLICE else [one(0)]
    total = 0
    for lo, got in zip(starts, parts):          # contiguous prefix actually read
        total += got
        if got < min(len(view), lo + _IO_SLICE) - lo:
            break
    return total


def _pwrite_full(fd: int, view: memoryview, offset: int) -> None:
    """Write `view` to `fd` at `offset` (regular file) in parallel slices."""
    import os

    def one(lo):
        hi, pos = min(len(view), lo + _IO_SLICE), lo
        while pos < hi:                              # os.pwrite may write less than asked
            k = os.pwrite(fd, view[pos:hi], offset + pos)
            if k <= 0:
                raise SystemExit(f"short write at byte {offset + pos}")
            pos += k
    if len(view) > _IO_SLICE:
        list(_io_pool().map(one, range(0, len(view), _IO_SLICE)))
    elif len(view):
        one(0)


def _seekable(f) -> bool:
    import os
    import stat
    try:
        return stat.S_ISREG(os.fstat(f.fileno()).st_mode)
    except (OSError, ValueError, AttributeError):
        return False


def main_sharded(a, rank: int, world: int) -> int:
    """One process per GPU (launched with `python -m torch.distributed.run --nproc-per-node N -m pythoncrt_amd.cli ...`):
    the clip's chunks of --batch frames are dealt round-robin over the ranks, each rank reads its chunks
    from the raw input file and writes them at the same offsets of the output file; with --persistence > 0 one
    float32 state frame per chunk boundary travels to the next rank (RCCL)."""
    import os
    import torch
    import torch.distributed as dist
    from .pipeline import FramePipeline, GpuShardEngine
    from .shard import FrameShard, ShardedRender
    from .text import make_text_overlay_rgba
    if a.input == "-" or not a.output or a.output == "-":
        raise SystemExit("the sharded CLI needs seekable --input and --output files")
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible; pythoncrt_amd has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    backend = os.environ.get("CRTFX_DIST_BACKEND", "nccl" if ndev >= world else "gloo")     # gloo: rehearsal of N ranks on fewer GPUs
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    rs = settings_from_args(a)
    fps_out = int(a.fps) if a.fps and a.fps > 0 else 24
    h, w = int(a.height), int(a.width)
    box = [a.noise_seed if a.noise_seed is not None else int.from_bytes(os.urandom(8), "little")]
    dist.broadcast_object_list(box, src=0)               # every rank draws the same grain stream
    overlay = make_text_overlay_rgba(w, h, a.text, a.text_font, a.text_size, a.text_color, (a.text_x, a.text_y)) if a.text else None
    pipe = FramePipeline(dev, h, w, rs, fps=fps_out, noise_seed=box[0], text_overlay_rgba=overlay, text_overlay_after=bool(a.text_after))
    B = max(1, int(a.batch))
    frame_bytes = h * w * 3
    n_frames = os.path.getsize(a.input) // frame_bytes
    shard = FrameShard(world, rank, B)
    # overlapped schedule where the chunk covers the IIR's settling time (shard.py): round r's state frame travels while round
    # r+1 is scanned, results come back one call late; any other case runs the synchronous protocol behind the same calls
    # — over gloo (rehearsals).  Over RCCL the synchronous hop stays the default until the overlapped one has run on a multi-GPU
    # box (it costs ~0.3 ms per round: one state frame over one xGMI link + the fix-up); CRTFX_SHARD_OVERLAP=1 opts in.
    overlap = backend == "gloo" or os.environ.get("CRTFX_SHARD_OVERLAP") == "1"
    render = ShardedRender(shard, rs.persistence, GpuShardEngine(pipe, B, slots=2), dist=dist, overlap=overlap)
    if rank == 0:
        with open(a.output, "wb") as f:
            f.truncate(n_frames * frame_bytes)
    dist.barrier()
    t0 = time.perf_counter()
    fin, fout = os.open(a.input, os.O_RDONLY), os.open(a.output, os.O_WRONLY)
    host = torch.empty((B, h, w, 3), dtype=torch.uint8).pin_memory()
    done = 0

    def commit(finished):
        nonlocal done
        for rr, out in finished:
            flo, fhi = shard.frame_range(rr, n_frames)
            _pwrite_full(fout, memoryview(out.cpu().numpy()).cast("B"), flo * frame_bytes)
            done += fhi - flo

    uploaded = torch.cuda.Event()
    for r in range(shard.rounds(n_frames)):
        lo, hi = shard.frame_range(r, n_frames)
        frames = None
        if hi > lo:
            uploaded.synchronize()                        # the previous chunk has left the pinned staging buffer
            view = memoryview(host.numpy()).cast("B")[: (hi - lo) * frame_bytes]
            if _pread_full(fin, view, lo * frame_bytes) != len(view):
                raise SystemExit(f"short read at frame {lo}")
            frames = host[: hi - lo].to(dev, non_blocking=True)
            uploaded.record()
        commit(render.submit_round(frames, r, active=shard.active_ranks(r, n_frames)))
    commit(render.close())                                # the round still in flight; frees the staged schedule's extra process groups
    os.close(fin); os.close(fout)
    dist.barrier()
    print(f"rank {rank}: {done} of {n_frames} frames, elapsed {time.perf_counter() - t0:.3f}s", file=sys.stderr)
    dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    import os as _os
    if int(_os.environ.get("WORLD_SIZE", "1")) > 1:
        if a.gui or not a.input or a.width <= 0 or a.height <= 0:
            raise SystemExit("pass --input, --width and --height")
        return main_sharded(a, int(_os.environ.get("RANK", "0")), int(_os.environ["WORLD_SIZE"]))
    if a.gui or not a.input:
        raise SystemExit("the GUI is not part of this path; pass --input (raw rgb24 file or '-')")
    if a.width <= 0 or a.height <= 0:
        raise SystemExit("raw rgb24 input needs --width and --height")
    import os
    import torch
    from .pipeline import FramePipeline
    from .text import make_text_overlay_rgba
    rs = settings_from_args(a)
    fps_out = int(a.fps) if a.fps and a.fps > 0 else 24            # ref:914
    h, w = int(a.height), int(a.width)
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible; pythoncrt_amd has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    seed = a.noise_seed if a.noise_seed is not None else int.from_bytes(os.urandom(8), "little")
    overlay = make_text_overlay_rgba(w, h, a.text, a.text_font, a.text_size, a.text_color, (a.text_x, a.text_y)) if a.text else None   # ref:1076
    pipe = FramePipeline(dev, h, w, rs, fps=fps_out, noise_seed=seed, text_overlay_rgba=overlay, text_overlay_after=bool(a.text_after))
    fin = sys.stdin.buffer if a.input == "-" else open(a.input, "rb", buffering=0)
    out_path = a.output if a.output else (a.input + "_crt.rgb" if a.input != "-" else "-")
    fout = sys.stdout.buffer if out_path == "-" else open(out_path, "wb")
    B = max(1, int(a.batch))
    frame_bytes = h * w * 3
    host_in = [torch.empty((B, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    host_out = [torch.empty((B, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    dev_in = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    dev_out = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    t0 = time.perf_counter()
    state, index, k = None, 0, 0
    pending = None                                                  # (slot, frames) of the batch still in flight

    # regular files: positional I/O on a few threads (a pipe / the terminal: the plain sequential calls)
    in_pos = _seekable(fin) and a.input != "-"
    out_pos = fout is not sys.stdout.buffer and _seekable(fout)
    in_off = out_off = 0

    def drain(p):
        nonlocal out_off
        slot, n = p
        done[slot].synchronize()
        view = memoryview(host_out[slot].numpy()).cast("B")[: n * frame_bytes]
        if out_pos:
            _pwrite_full(fout.fileno(), view, out_off)
            out_off += len(view)
        else:
            fout.write(view)

    while True:
        slot = k & 1
        if in_pos:
            got = _pread_full(fin.fileno(), memoryview(host_in[slot].numpy()).cast("B"), in_off)
            in_off += got
        else:
            got = _read_into(fin, memoryview(host_in[slot].numpy()).cast("B"))
        n = got // frame_bytes                                      # a trailing partial frame is dropped, as ffmpeg's rawvideo demuxer does
        if n:
            dev_in[slot][:n].copy_(host_in[slot][:n], non_blocking=True)
            _, state = pipe.run(dev_in[slot][:n], first_index=index, state=state, out=dev_out[slot][:n])
            host_out[slot][:n].copy_(dev_out[slot][:n], non_blocking=True)
            done[slot].record()
        if pending is not None:
            drain(pending)
        pending = (slot, n) if n else None
        index += n
        k += 1
        if n < B:
            break
    if pending is not None:
        drain(pending)
    fout.flush()
    if fout is not sys.stdout.buffer:
        fout.close()
    if fin is not sys.stdin.buffer:
        fin.close()
    print(f"{index} frames, elapsed {time.perf_counter() - t0:.3f}s", file=sys.stderr)      # ref:1269
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
