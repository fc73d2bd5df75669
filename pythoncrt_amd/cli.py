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

Host staging (the reference's reader / writer pipes ref:469-514, :1003-1014, :1101): a reader thread fills pinned
input batches, a writer thread drains pinned output batches, and the GPU side runs on THREE streams — upload, kernels, download —
chained by events, so that batch k+1's upload, batch k's kernels and batch k-1's download are in flight together (the two PCIe
directions are independent DMA engines) while the host reads batch k+2 and writes batch k-2.

Regular files skip the pinned staging where the platform lets them (`--io auto`, round 5): the input file is mapped MAP_SHARED and the
windows of the mapping a batch covers are registered with the HIP runtime (hipHostRegister), so the upload DMA reads the page cache itself —
no page-cache -> pinned memcpy; the output file is sized up front (ftruncate), mapped and registered the same way and the download DMA writes
the page cache itself — no pinned -> page-cache copy.  Whether that wins depends on the filesystem (profiles/r05_hostreg_probe*.txt: on a
disk-backed filesystem registering a fresh 398 MB batch takes 4.6 ms, on tmpfs — 4 KB pages, every one marked accessed on first touch —
23-30 ms, slower than sixteen memcpy threads), so the first batch is timed and the staged path takes over when registration is refused or
slower than `_MAPPED_MIN_GBS`.  Pipes get the largest pipe buffer the kernel allows (F_SETPIPE_SZ, 1 MiB) and are read / written straight
from / to the pinned slots.
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
    p.add_argument("--staging-report", action="store_true", help="print where the reader / GPU-feeding / writer threads spent their time")
    p.add_argument("--noise-seed", type=int, default=None, help="seed of the counter-based grain RNG (default: random)")
    p.add_argument("--io", type=str, default="staged", choices=["auto", "staged", "mapped"],
                   help="regular files: through pinned staging slots (staged, the default: at the PCIe rate on a disk-backed filesystem), DMA "
                        "straight from / into the registered file mapping (mapped: measured slower on this platform, profiles/r05_cli_throughput.txt), "
                        "or mapped where the first batch shows it is faster (auto)")
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
_IO_SLICE = 8 << 20             # bytes per positional read / write / copy call


def _io_pool():
    """Threads for positional file I/O: one os.preadv / os.pwrite moves ~8 GB/s out of / into the page cache (a memcpy on one core,
    GIL released), a 4K frame is 24.9 MB each way — the raw rgb24 edge, not the GPU, would set the CLI's frame rate."""
    global _IO_POOL
    if _IO_POOL is None:
        import concurrent.futures
        import os
        try:
            cpus = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cpus = os.cpu_count() or 2
        want = int(os.environ.get("CRTFX_IO_THREADS", "0") or 0)          # A/B knob (tools/cli_throughput.sh); default: half the cpus, at most 16
        _IO_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=want if want > 0 else max(1, min(16, cpus // 2)), thread_name_prefix="crtfx-io")
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


_LIBC = None


def _madvise(addr: int, length: int, advice: int) -> int:
    """madvise(2) through ctypes: the call runs WITHOUT the GIL (mmap.madvise of CPython 3.10 keeps it), so the zapper thread's drops of a
    400 MB batch (~13 ms) never hold up the Python threads that feed the GPU."""
    global _LIBC
    if _LIBC is None:
        import ctypes
        _LIBC = ctypes.CDLL(None, use_errno=True)
        _LIBC.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _LIBC.madvise.restype = ctypes.c_int
    return _LIBC.madvise(addr, length, advice)


class _MappedInput:
    """A regular input file mapped read-only: a batch is copied out of the page cache by user-space memcpy on the I/O threads
    (numpy's copy loop releases the GIL) instead of read(2).  Measured on the GPU box, 1.6 GB from tmpfs into a pinned buffer, eight
    threads (tools/io_rates.py): 76 GB/s — against 18 GB/s for os.preadv on the FIRST read of a freshly written file (61 GB/s on later
    reads), which is what capped the CLI at ~700 4K frames/s whatever the GPU side did.  MADV_WILLNEED on the batch after next starts
    the readahead of a file that is not in the page cache yet."""

    def __init__(self, fd: int):
        import mmap
        import os
        self.fd, self._os = fd, os
        self.size = os.fstat(fd).st_size
        self.map = mmap.mmap(fd, self.size, prot=mmap.PROT_READ) if self.size > 0 else None
        self.arr = np.frombuffer(self.map, dtype=np.uint8) if self.map is not None else None
        self.base = int(self.arr.ctypes.data) if self.arr is not None else 0
        self._mmap = mmap
        # the page-table entries of what has been copied are dropped (the pages stay in the page cache) by ONE thread of its own, batch by
        # batch behind the copies: unmapping costs ~0.15 us per 4 KB page wherever it is paid — in one piece at exit 0.9 s for a 24 GB clip;
        # from the sixteen copy threads, slice by slice, the drops contend in the kernel (tmpfs: 28 ms per 400 MB batch when they really run in
        # parallel, 17 ms serialised by the GIL as in round 4) and sit on the reader's critical path; one thread beside the copies does
        # a batch in ~13 ms and delays nobody (profiles/r05_cli_throughput.txt)
        import queue
        import threading
        self._zap_q = queue.Queue()
        self._zap_mode = os.environ.get("CRTFX_IO_DONTNEED", "thread")      # A/B knob: "thread" (default) | "0" (never: left to process exit) | "slice" (on the copy threads, round 4)
        self._zap_thr = None
        self._zap_err = False
        if self.map is not None and self._zap_mode == "thread" and hasattr(mmap, "MADV_DONTNEED"):
            def zapper():
                while True:
                    job = self._zap_q.get()
                    if job is None:
                        return
                    a0, a1 = job
                    while a0 < a1 and not self._zap_err:
                        k = min(a1 - a0, 32 << 20)
                        if _madvise(self.base + a0, k, mmap.MADV_DONTNEED) != 0:
                            self._zap_err = True      # the first failure ends the drops (they are an optimisation): a raw address is never advised blindly
                        a0 += k
            self._zap_thr = threading.Thread(target=zapper, name="crtfx-zap", daemon=True)
            self._zap_thr.start()

    def read_into(self, dst: np.ndarray, offset: int) -> int:
        """Fill the uint8 array `dst` from the file at `offset`; returns the bytes copied (short only at end of file)."""
        # a file that was truncated under us ends the clip early (a short read, as read(2) would report it) instead of a SIGBUS from the
        # pages that are gone: the size is looked at again before every batch
        try:
            now = self._os.fstat(self.fd).st_size
        except OSError:
            now = self.size
        n = max(0, min(dst.size, min(self.size, now) - offset))
        if n <= 0:
            return 0
        ahead = min(self.size, offset + 3 * dst.size) - (offset + n)
        if ahead > 0 and hasattr(self._mmap, "MADV_WILLNEED"):
            page = self._mmap.PAGESIZE
            lo = ((offset + n) // page) * page
            _madvise(self.base + lo, min(self.size - lo, ahead + page), self._mmap.MADV_WILLNEED)

        page = self._mmap.PAGESIZE
        drop = getattr(self._mmap, "MADV_DONTNEED", None) if self._zap_mode == "slice" else None

        def one(lo):
            hi = min(n, lo + _IO_SLICE)
            np.copyto(dst[lo:hi], self.arr[offset + lo:offset + hi])
            # CRTFX_IO_DONTNEED=slice (round 4's form, kept for the A/B): drop the page-table entries of what this thread has copied, slice by
            # slice on the I/O threads; the default hands the whole batch to the zapper thread below instead
            if drop is not None:
                a0, a1 = ((offset + lo + page - 1) // page) * page, ((offset + hi) // page) * page
                if a1 > a0:
                    try:
                        self.map.madvise(drop, a0, a1 - a0)      # mmap.madvise keeps the GIL: the drops of the sixteen threads run one after the other (on purpose, see __init__)
                    except (OSError, ValueError):
                        pass
        if n > _IO_SLICE:
            list(_io_pool().map(one, range(0, n, _IO_SLICE)))
        else:
            one(0)
        if self._zap_thr is not None:
            a0, a1 = ((offset + page - 1) // page) * page, ((offset + n) // page) * page
            if a1 > a0:
                self._zap_q.put((a0, a1))
        return n

    def close(self):
        if self._zap_thr is not None:
            self._zap_q.put(None)
            self._zap_thr.join()                    # no timeout: the thread advises RAW addresses of this mapping, which must outlive it (a drop of a
            self._zap_thr = None                    # 400 MB batch takes ~13 ms; the queue holds at most the batches read)
        self.arr = None
        if self.map is not None:
            try:
                self.map.close()
            except BufferError:         # a view of the map is still referenced somewhere: the mapping goes with the process
                pass
            self.map = None


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


def _grow_pipe(f, want: int = 0) -> int:
    """F_SETPIPE_SZ on a pipe end (Linux): the default 64 KiB buffer makes a 24.9 MB 4K frame 380 wake-ups of the process on the other
    side; 1 MiB is what an unprivileged process may ask for (/proc/sys/fs/pipe-max-size).  Returns the buffer size now in force (0: not a
    pipe / not supported)."""
    import fcntl
    import os
    import stat
    try:
        fd = f.fileno()
        if not stat.S_ISFIFO(os.fstat(fd).st_mode):
            return 0
        if want <= 0:                              # CRTFX_PIPE_SIZE: A/B of the buffer size (tools/cli_throughput.sh); 0 bytes = leave the pipe as it is
            want = int(os.environ.get("CRTFX_PIPE_SIZE", 1 << 20))
        F_SETPIPE_SZ, F_GETPIPE_SZ = getattr(fcntl, "F_SETPIPE_SZ", 1031), getattr(fcntl, "F_GETPIPE_SZ", 1032)
        try:
            if want > 0:
                fcntl.fcntl(fd, F_SETPIPE_SZ, want)
        except OSError:
            pass                                   # above pipe-max-size: keep what we have
        return int(fcntl.fcntl(fd, F_GETPIPE_SZ))
    except (OSError, ValueError, AttributeError):
        return 0


class _HostDma:
    """hipHostRegister / hipHostUnregister / hipMemcpyAsync of the HIP runtime this process has already loaded (torch's), through ctypes —
    plain pointers and sizes.  What they are for: the PCIe DMA engines read a batch straight out of, or write it straight into, a shared
    mapping of the raw rgb24 FILE (the page cache), with no pinned staging copy on the host."""
    H2D, D2H = 1, 2
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes
            path = None
            try:
                with open("/proc/self/maps") as maps:
                    for line in maps:
                        if "libamdhip64" in line:
                            path = line.split()[-1]
                            break
            except OSError:
                pass
            lib = ctypes.CDLL(path or "libamdhip64.so")
            lib.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
            lib.hipHostRegister.restype = ctypes.c_int
            lib.hipHostUnregister.argtypes = [ctypes.c_void_p]
            lib.hipHostUnregister.restype = ctypes.c_int
            lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
            lib.hipMemcpyAsync.restype = ctypes.c_int
            lib.hipGetLastError.restype = ctypes.c_int
            cls._lib = lib
        return cls._lib

    @classmethod
    def register(cls, ptr: int, n: int) -> bool:
        lib = cls.lib()
        if lib.hipHostRegister(ptr, n, 0) != 0:
            lib.hipGetLastError()                  # clear the sticky error: the caller falls back to staging
            return False
        return True

    @classmethod
    def unregister(cls, ptr: int):
        if cls.lib().hipHostUnregister(ptr) != 0:
            cls.lib().hipGetLastError()

    @classmethod
    def copy(cls, dst: int, src: int, n: int, kind: int, stream: int):
        rc = cls.lib().hipMemcpyAsync(dst, src, n, kind, stream)
        if rc != 0:
            raise RuntimeError(f"hipMemcpyAsync failed with error {rc}")


_MAPPED_MIN_GBS = 30.0          # --io auto: registration slower than this per byte (tmpfs: 13-17 GB/s) loses to the memcpy staging


class _RegisteredMap:
    """A MAP_SHARED mapping of a regular file whose pages the GPU's DMA engines address directly.  Windows of WIN bytes (page multiples,
    on a fixed grid so that the page two neighbouring batches share is never registered twice) are registered with the HIP runtime when a
    batch first needs them and unregistered when the last batch using them is done."""
    WIN = 32 << 20

    def __init__(self, fd: int, size: int, writable: bool):
        import mmap
        import threading
        self.size = int(size)
        self.map = mmap.mmap(fd, self.size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ | (mmap.PROT_WRITE if writable else 0))
        self.arr = np.frombuffer(self.map, dtype=np.uint8)
        self.base = int(self.arr.ctypes.data)
        self.maplen = -(-self.size // mmap.PAGESIZE) * mmap.PAGESIZE
        self.ref, self.lock = {}, threading.Lock()
        self.t_reg = 0.0                        # seconds spent in hipHostRegister

    def _span(self, w):
        lo = w * self.WIN
        return lo, min(self.WIN, self.maplen - lo)

    def acquire(self, off: int, n: int) -> bool:
        """Make the bytes [off, off + n) DMA-addressable; False when the runtime refuses (nothing stays registered for this call)."""
        ws = range(off // self.WIN, (off + n - 1) // self.WIN + 1)
        done = []
        with self.lock:
            for w in ws:
                if self.ref.get(w, 0) == 0:
                    lo, ln = self._span(w)
                    t = time.perf_counter()
                    ok = _HostDma.register(self.base + lo, ln)
                    self.t_reg += time.perf_counter() - t
                    if not ok:
                        for v in done:
                            self._drop(v)
                        return False
                self.ref[w] = self.ref.get(w, 0) + 1
                done.append(w)
        return True

    def _drop(self, w):
        c = self.ref.get(w, 0) - 1
        if c <= 0:
            self.ref.pop(w, None)
            _HostDma.unregister(self.base + w * self.WIN)
        else:
            self.ref[w] = c

    def release(self, off: int, n: int):
        with self.lock:
            for w in range(off // self.WIN, (off + n - 1) // self.WIN + 1):
                self._drop(w)

    def copy(self, kind: int, dev_ptr: int, off: int, n: int, stream: int):
        """hipMemcpyAsync between device memory at dev_ptr and the file bytes [off, off + n), window by window: every window is a
        registration of its own, and the runtime rejects a single copy that runs across two of them."""
        pos = 0
        while pos < n:
            w = (off + pos) // self.WIN
            k = min(n - pos, (w + 1) * self.WIN - (off + pos))
            host = self.base + off + pos
            if kind == _HostDma.H2D:
                _HostDma.copy(dev_ptr + pos, host, k, kind, stream)
            else:
                _HostDma.copy(host, dev_ptr + pos, k, kind, stream)
            pos += k

    def close(self):
        with self.lock:
            for w in list(self.ref):
                _HostDma.unregister(self.base + w * self.WIN)
            self.ref.clear()
        self.arr = None
        try:
            self.map.close()
        except (BufferError, ValueError):
            pass


class _MapTok:
    """One batch that lives in a registered file mapping: byte range + the flow-control slot it holds."""
    __slots__ = ("off", "nbytes", "slot")

    def __init__(self, off, nbytes, slot):
        self.off, self.nbytes, self.slot = off, nbytes, slot


class _Reader:
    """A thread that gets input batches ready ahead of the GPU.  `jobs` yields (byte offset | None, frames) requests — an offset for
    positional reads of a regular file, None for the next bytes of a stream — and `get()` hands back (token, frames, bytes) in order, or
    None at the end; `upload(token, frames, dst, stream)` enqueues the batch's host-to-device copy and `release(token)` returns its slot
    once that copy has completed.  Two ways a batch gets ready:
      staged   filled into a pinned (B, H, W, 3) uint8 slot (token = slot index): read(2) / readinto for a stream, memcpy out of a
               private mapping on the I/O threads for a regular file;
      mapped   (io = "mapped" / "auto", regular files) the windows of a SHARED mapping of the file that the batch covers are registered
               with the HIP runtime and the upload reads the page cache itself (token = _MapTok).  "auto" times the first batch and goes
               back to staging when registration is refused or runs below _MAPPED_MIN_GBS."""

    def __init__(self, fin, positional: bool, jobs, shape, frame_bytes: int, slots: int = 3, io: str = "staged", autostart: bool = True):
        import os
        import queue
        import threading
        import torch
        self.fin, self.positional, self.frame_bytes, self.shape = fin, positional, frame_bytes, shape
        self._torch = torch
        self._bufs = [None] * slots
        self.free, self.full = queue.Queue(), queue.Queue()      # the free-slot queue already bounds what is in flight
        for i in range(slots):
            self.free.put(i)
        self.err = None
        self.t_wait = self.t_io = 0.0          # seconds this thread waited for a free slot / spent reading (the --staging-report line)
        self.mapped = None                     # private mapping the staged path copies out of
        self.rmap = None                       # shared, registered mapping of the mapped path
        self.mode = "staged"                   # what the NEXT batch will use
        self.mapped_batches = self.staged_batches = 0
        self.note = ""
        if positional:
            try:
                self.mapped = _MappedInput(fin.fileno())
                if self.mapped.map is None:
                    self.mapped = None
            except (OSError, ValueError):
                self.mapped = None             # not mappable: positional read(2) calls instead
            if io in ("mapped", "auto") and self.mapped is not None:
                try:
                    self.rmap = _RegisteredMap(fin.fileno(), self.mapped.size, writable=False)
                    self.mode = "mapped"
                except (OSError, ValueError, AttributeError) as e:
                    self.note = f"input mapping refused ({e.__class__.__name__}): staged"
        self._pin_lock = threading.Lock()
        if self.mode == "staged":
            # pinned up front, as the GPU loop expects when it starts.  (Six 16-frame 4K slots take 0.5 s of hipHostMalloc; pinning all but the
            # first on a helper thread beside the first batches was tried in round 5: the pipeline then ran 0.25 s longer — the runtime
            # serialises the allocations with the copies' enqueues — and the run as a whole no shorter.)
            for i in range(slots):
                self._buf(i)

        def stage(i, off, nfr):
            view = memoryview(self._buf(i).numpy()).cast("B")[: nfr * frame_bytes]
            if self.mapped is not None:
                got = self.mapped.read_into(self._buf(i).numpy().reshape(-1)[: nfr * frame_bytes], off)
            else:
                got = _pread_full(fin.fileno(), view, off) if positional else _read_into(fin, view)
            self.staged_batches += 1
            return i, got, len(view)

        def map_batch(i, off, nfr):
            """-> (token, got, wanted) or None when this batch has to be staged"""
            try:
                now = os.fstat(fin.fileno()).st_size
            except OSError:
                now = self.rmap.size
            want = nfr * frame_bytes
            got = max(0, min(want, min(self.rmap.size, now) - off))
            nbytes = (got // frame_bytes) * frame_bytes
            if nbytes:
                t = time.perf_counter()
                ok = self.rmap.acquire(off, nbytes)
                dt = time.perf_counter() - t
                if not ok:
                    self.mode, self.note = "staged", "hipHostRegister refused the input mapping: staged"
                    return None
                if io == "auto" and self.mapped_batches == 0 and nbytes / max(dt, 1e-9) / 1e9 < _MAPPED_MIN_GBS:
                    # this batch is registered and is used as it is; the ones behind it are staged
                    self.mode = "staged"
                    self.note = f"registering the input mapping ran at {nbytes / max(dt, 1e-9) / 1e9:.1f} GB/s (< {_MAPPED_MIN_GBS:.0f}): staged from the second batch on"
            self.mapped_batches += 1
            return _MapTok(off, nbytes, i), got, want

        def loop():
            try:
                for off, nfr in jobs:
                    t = time.perf_counter()
                    i = self.free.get()
                    self.t_wait += time.perf_counter() - t
                    if i is None:
                        return
                    t = time.perf_counter()
                    res = map_batch(i, off, nfr) if self.mode == "mapped" else None
                    if res is None:
                        res = stage(i, off, nfr)
                    tok, got, want = res
                    self.t_io += time.perf_counter() - t
                    self.full.put((tok, got // frame_bytes, got))
                    if got < want:
                        break
            except BaseException as e:      # noqa: BLE001 - re-raised on the consumer's side
                self.err = e
            self.full.put(None)
        self.thr = threading.Thread(target=loop, name="crtfx-reader", daemon=True)
        if autostart:
            self.thr.start()

    def start(self):
        """autostart=False: the first read is issued here (the CLI times its pipeline from this point, with every staging slot already pinned)."""
        self.thr.start()

    def _buf(self, i):
        if self._bufs[i] is None:
            with self._pin_lock:
                if self._bufs[i] is None:
                    self._bufs[i] = self._torch.empty(self.shape, dtype=self._torch.uint8).pin_memory()
        return self._bufs[i]

    @property
    def bufs(self):
        return [self._buf(i) for i in range(len(self._bufs))]

    def get(self):
        item = self.full.get()
        if self.err is not None:
            raise self.err
        return item

    def upload(self, tok, n: int, dst, stream):
        """Enqueue the host-to-device copy of the batch's first n frames into dst[:n] on `stream` (a torch stream)."""
        if isinstance(tok, _MapTok):
            self.rmap.copy(_HostDma.H2D, dst.data_ptr(), tok.off, n * self.frame_bytes, stream.cuda_stream)
        else:
            with self._torch.cuda.stream(stream):
                dst[:n].copy_(self._buf(tok)[:n], non_blocking=True)

    def release(self, tok):
        if isinstance(tok, _MapTok):
            if tok.nbytes:
                self.rmap.release(tok.off, tok.nbytes)
            self.free.put(tok.slot)
        else:
            self.free.put(tok)

    def close(self):
        self.free.put(None)
        self.thr.join(timeout=30)
        if self.rmap is not None:
            self.rmap.close()
        if self.mapped is not None:
            self.mapped.close()


class _Writer:
    """A thread that finishes output batches behind the GPU.  `slot()` blocks until a destination is free and returns its token;
    `download(token, frames, src, stream)` enqueues the device-to-host copy; `put(token, frames, event, offset)` queues the batch for the
    thread, which waits for the event and
      staged   writes the pinned slot out (token = slot index; positional pwrite slices for a regular file, write() for a stream);
      mapped   (regular output file whose final size is known: `plan` = [(offset, bytes), ...]) has nothing left to write — the file was
               sized with ftruncate, mapped MAP_SHARED, and a helper thread registered each batch's windows with the HIP runtime ahead of
               the GPU, so the download DMA wrote the page cache itself; the thread only unregisters the windows.
    After the first write error nothing more is written: the remaining jobs are drained (their slots go back) and the first error is kept."""

    def __init__(self, fout, positional: bool, shape, frame_bytes: int, slots: int = 3, plan=None, io: str = "staged"):
        import os
        import queue
        import threading
        import torch
        self._torch, self.shape, self.frame_bytes = torch, shape, frame_bytes
        self._os, self._closed = os, False
        try:
            self._fd = fout.fileno() if positional else None
        except (OSError, ValueError, AttributeError):
            self._fd = None
        self._bufs = [None] * slots
        self._pin_lock = threading.Lock()
        self.free, self.work = queue.Queue(), queue.Queue()
        for i in range(slots):
            self.free.put(i)
        self.err, self.frames = None, 0
        self.t_wait = self.t_io = 0.0          # seconds this thread waited for downloads / spent writing
        self.rmap, self.ready, self.prep = None, None, None
        self.note = ""
        self.mapped_batches = 0
        self.t_prep = 0.0
        if positional and plan and io in ("mapped", "auto"):
            total = plan[-1][0] + plan[-1][1]
            try:
                os.ftruncate(fout.fileno(), total)
                self.rmap = _RegisteredMap(fout.fileno(), total, writable=True)
            except (OSError, ValueError, AttributeError) as e:
                self.rmap, self.note = None, f"output mapping refused ({e.__class__.__name__}): staged"
        if self.rmap is not None:
            self.ready = queue.Queue()
            self.tokens = threading.Semaphore(slots)      # batches registered ahead of / in flight on the GPU

            def prepare():
                for off, nbytes in plan:
                    self.tokens.acquire()
                    if self.stop_prep:
                        return
                    t = time.perf_counter()
                    ok = self.rmap.acquire(off, nbytes)
                    self.t_prep += time.perf_counter() - t
                    if not ok:
                        self.note = "hipHostRegister refused the output mapping: staged"
                        self.ready.put(None)           # from here on: pinned slots + pwrite into the (already sized) file
                        return
                    self.ready.put(_MapTok(off, nbytes, None))
                self.ready.put(None)
            self.stop_prep = False
            self.prep = threading.Thread(target=prepare, name="crtfx-out-prep", daemon=True)
            self.prep.start()
        else:
            for i in range(slots):
                self._buf(i)

        def loop():
            while True:
                job = self.work.get()
                if job is None:
                    return
                tok, n, ev, off = job
                try:
                    t = time.perf_counter()
                    ev.synchronize()
                    self.t_wait += time.perf_counter() - t
                    if isinstance(tok, _MapTok):
                        self.frames += n
                    elif self.err is None:          # after a write error: no further writes, the slots still go back
                        view = memoryview(self._buf(tok).numpy()).cast("B")[: n * frame_bytes]
                        t = time.perf_counter()
                        if positional:
                            _pwrite_full(fout.fileno(), view, off)
                        else:
                            fout.write(view)
                        self.t_io += time.perf_counter() - t
                        self.frames += n
                except BaseException as e:      # noqa: BLE001
                    if self.err is None:
                        self.err = e
                if isinstance(tok, _MapTok):
                    self.rmap.release(tok.off, tok.nbytes)
                    self.tokens.release()
                else:
                    self.free.put(tok)
        self.thr = threading.Thread(target=loop, name="crtfx-writer", daemon=True)
        self.thr.start()

    def _buf(self, i):
        if self._bufs[i] is None:
            with self._pin_lock:
                if self._bufs[i] is None:
                    self._bufs[i] = self._torch.empty(self.shape, dtype=self._torch.uint8).pin_memory()
        return self._bufs[i]

    @property
    def bufs(self):
        return [self._buf(i) for i in range(len(self._bufs))]

    def slot(self):
        if self.err is not None:
            raise self.err
        if self.ready is not None:
            tok = self.ready.get()
            if tok is not None:
                self.mapped_batches += 1
                return tok
            self.ready = None                      # the plan is used up (a longer input than planned cannot happen) or registration failed
        return self.free.get()

    def download(self, tok, n: int, src, stream):
        """Enqueue the device-to-host copy of src[:n] into the batch's destination on `stream` (a torch stream)."""
        if isinstance(tok, _MapTok):
            self.rmap.copy(_HostDma.D2H, src.data_ptr(), tok.off, n * self.frame_bytes, stream.cuda_stream)
        else:
            with self._torch.cuda.stream(stream):
                self._buf(tok)[:n].copy_(src[:n], non_blocking=True)

    def put(self, tok, n: int, ev, off):
        self.work.put((tok, n, ev, off))

    def close(self):
        self.work.put(None)
        self.thr.join()
        if self.prep is not None:
            self.stop_prep = True
            for _ in range(len(self._bufs) + 1):
                self.tokens.release()
            self.prep.join(timeout=30)
        if self.rmap is not None:
            if self.mapped_batches and not self._closed:
                # the download DMA wrote MAP_SHARED pages of the output file directly: msync + fsync before the windows are unregistered and the
                # mapping goes, so that the frames' way to the disk does not hang on how the driver marked the pinned pages (INTEGRATION.md,
                # "--io mapped durability"); on tmpfs both calls return at once
                try:
                    self.rmap.map.flush()
                    if self._fd is not None:
                        self._os.fsync(self._fd)
                except (OSError, ValueError) as e:
                    if self.err is None:
                        self.err = e
            self.rmap.close()
        self._closed = True
        if self.err is not None:
            raise self.err


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
    if "MASTER_PORT" not in os.environ:                  # CRTFX_FORCE_DIST=1 without a launcher
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
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
    ov = os.environ.get("CRTFX_SHARD_OVERLAP")            # "1" / "0": force the overlapped / the synchronous schedule (tests run both over gloo)
    overlap = (ov == "1") if ov in ("0", "1") else backend == "gloo"
    # test hook: hold every download back by this many milliseconds of GPU time, so that a missing ordering between a round's download and the
    # next round's kernels shows as wrong bytes instead of passing by timing (tests/test_cli_gpu.py::test_sharded_cli_synchronous_schedule)
    down_delay_ms = float(os.environ.get("CRTFX_TEST_DOWNLOAD_DELAY_MS", "0") or 0)
    # three output slots: round r's frames may still be on their way to the host (download stream) while round r + 1 is scanned and —
    # overlapped schedule, results one call late — round r + 2 is enqueued
    # world 1 arrives here only as CRTFX_FORCE_DIST=1 (main): the one-rank ring — the same protocol with rank 0 as its own neighbour, so that
    # this function's RCCL branch (eager communicator, object broadcast, the hop between the three streams below) runs on a one-GPU box
    render = ShardedRender(shard, rs.persistence, GpuShardEngine(pipe, B, slots=3), dist=dist, overlap=overlap, loopback=(world == 1))
    if rank == 0:
        with open(a.output, "wb") as f:
            f.truncate(n_frames * frame_bytes)
    dist.barrier()
    t0 = time.perf_counter()
    fin, fout = open(a.input, "rb", buffering=0), open(a.output, "r+b", buffering=0)
    n_rounds = shard.rounds(n_frames)
    mine = [(r,) + shard.frame_range(r, n_frames) for r in range(n_rounds)]
    reader = _Reader(fin, True, ((lo * frame_bytes, hi - lo) for _, lo, hi in mine if hi > lo), (B, h, w, 3), frame_bytes, slots=2)
    writer = _Writer(fout, True, (B, h, w, 3), frame_bytes, slots=2)
    compute = torch.cuda.current_stream(dev)
    s_up, s_down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    dev_in = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    in_free = [None, None]                                # per device input slot: the event after which the kernels no longer read it
    downs = []                                            # download events in order of issue

    def commit(finished):
        for rr, out in finished:
            flo, fhi = shard.frame_range(rr, n_frames)
            n = fhi - flo
            i = writer.slot()
            ready = torch.cuda.Event()
            ready.record(compute)                         # the round's kernels (and fix-up) are enqueued behind this point at the latest
            s_down.wait_event(ready)
            with torch.cuda.stream(s_down):
                if down_delay_ms > 0:
                    torch.cuda._sleep(int(down_delay_ms * 2.0e6))      # ~2 GHz shader clock; the exact length does not matter
                writer.bufs[i][:n].copy_(out[:n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(s_down)
            downs.append(ev)
            writer.put(i, n, ev, flo * frame_bytes)

    k, pend = 0, None
    for r, lo, hi in mine:
        frames = None
        if hi > lo:
            item = reader.get()
            if item is None or item[1] != hi - lo:
                raise SystemExit(f"short read at frame {lo}")
            i, n, _ = item
            d = k & 1
            if in_free[d] is not None:
                s_up.wait_event(in_free[d])               # the kernels that read this device slot two rounds ago are done
            with torch.cuda.stream(s_up):
                dev_in[d][:n].copy_(reader.bufs[i][:n], non_blocking=True)
                up = torch.cuda.Event()
                up.record(s_up)
            compute.wait_event(up)
            frames = dev_in[d][:n]
            k += 1
        # the engine's three output slots come round every third round: every download but the latest must have left them
        for ev in downs[:-1]:
            compute.wait_event(ev)
        del downs[:-1]
        commit(render.submit_round(frames, r, active=shard.active_ranks(r, n_frames)))
        if pend is not None:                              # the previous chunk's pinned slot: free again once its upload has completed (long done)
            pend[0].synchronize()
            reader.release(pend[1])
            pend = None
        if hi > lo:
            pend = (up, i)
            ev = torch.cuda.Event()
            ev.record(compute)                            # the scan that reads dev_in[d] is enqueued by now (the fix-up reads local states only)
            in_free[d] = ev
    if pend is not None:
        pend[0].synchronize()
        reader.release(pend[1])
    commit(render.close())                                # the round still in flight; frees the staged schedule's extra process groups
    writer.close()
    reader.close()
    done = writer.frames
    fin.close(); fout.close()
    dist.barrier()
    print(f"rank {rank}: {done} of {n_frames} frames, elapsed {time.perf_counter() - t0:.3f}s", file=sys.stderr)
    dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    import os as _os
    if int(_os.environ.get("WORLD_SIZE", "1")) > 1 or _os.environ.get("CRTFX_FORCE_DIST") == "1":
        if a.gui or not a.input or a.width <= 0 or a.height <= 0:
            raise SystemExit("pass --input, --width and --height")
        return main_sharded(a, int(_os.environ.get("RANK", "0")), int(_os.environ.get("WORLD_SIZE", "1")))
    if a.gui or not a.input:
        raise SystemExit("the GUI is not part of this path; pass --input (raw rgb24 file or '-')")
    if a.width <= 0 or a.height <= 0:
        raise SystemExit("raw rgb24 input needs --width and --height")
    import os
    t_start = time.perf_counter()
    import torch
    from .pipeline import FramePipeline
    from .text import make_text_overlay_rgba
    t_imports = time.perf_counter() - t_start
    rs = settings_from_args(a)
    fps_out = int(a.fps) if a.fps and a.fps > 0 else 24            # ref:914
    h, w = int(a.height), int(a.width)
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible; pythoncrt_amd has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    seed = a.noise_seed if a.noise_seed is not None else int.from_bytes(os.urandom(8), "little")
    overlay = make_text_overlay_rgba(w, h, a.text, a.text_font, a.text_size, a.text_color, (a.text_x, a.text_y)) if a.text else None   # ref:1076
    pipe = FramePipeline(dev, h, w, rs, fps=fps_out, noise_seed=seed, text_overlay_rgba=overlay, text_overlay_after=bool(a.text_after))
    t_engine = time.perf_counter() - t_start - t_imports
    fin = sys.stdin.buffer if a.input == "-" else open(a.input, "rb", buffering=0)
    out_path = a.output if a.output else (a.input + "_crt.rgb" if a.input != "-" else "-")
    B = max(1, int(a.batch))
    frame_bytes = h * w * 3
    t0 = time.perf_counter()
    # regular files: positional I/O on a few threads (a pipe / the terminal: the plain sequential calls)
    in_pos = _seekable(fin) and a.input != "-"
    # the output is opened with read access too (a MAP_SHARED, PROT_WRITE mapping needs O_RDWR).  --io mapped | auto only: an existing regular
    # output file behind a regular input file is NOT truncated at open — it is sized to the clip below and the pages it already has in the page
    # cache are overwritten in place (registering them is several times faster than allocating new ones).  The default (--io staged) truncates at
    # open and the file GROWS as batches are written, so that a render that dies leaves a short file, never a full-length one holding frames of an
    # earlier render (round-5 advisor finding); the sized paths cut the file back to the frames actually written when the render raises (below).
    if in_pos and out_path != "-" and os.path.exists(out_path) and os.path.samefile(out_path, a.input):
        raise SystemExit("--output is the input file")
    presize = in_pos and out_path != "-" and a.io in ("mapped", "auto")
    keep_pages = presize and os.path.isfile(out_path)
    fout = sys.stdout.buffer if out_path == "-" else open(out_path, "r+b" if keep_pages else "w+b")
    out_pos = fout is not sys.stdout.buffer and _seekable(fout)
    pipe_in, pipe_out = (0 if in_pos else _grow_pipe(fin)), (0 if out_pos else _grow_pipe(fout))      # pipes: the largest buffer the kernel allows
    # a regular input file fixes the clip's length, and with it the output file's size: the output can then be sized, mapped and registered
    # up front (the zero-copy download); a stream's length is unknown, its output goes through the pinned slots
    out_plan = None
    if in_pos and out_pos:
        n_total = os.fstat(fin.fileno()).st_size // frame_bytes
        out_plan = [(k * B * frame_bytes, min(B, n_total - k * B) * frame_bytes) for k in range((n_total + B - 1) // B)] or None
        if presize and out_pos:
            os.ftruncate(fout.fileno(), n_total * frame_bytes)

    def jobs():                                                     # whole batches until the stream ends (the reader stops at a short read)
        off = 0
        while True:
            yield (off if in_pos else None), B
            off += B * frame_bytes
    NS = 3
    reader = _Reader(fin, in_pos, jobs(), (B, h, w, 3), frame_bytes, slots=NS, io=a.io, autostart=False)
    writer = _Writer(fout, out_pos, (B, h, w, 3), frame_bytes, slots=NS, plan=out_plan, io=a.io)
    t_pipe = time.perf_counter()                                    # the pipeline proper: first read issued ... last batch written (the --staging-report line)
    t_slots = t_pipe - t_start - t_imports - t_engine
    reader.start()
    dev_in = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(NS)]
    dev_out = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(NS)]
    compute = torch.cuda.current_stream(dev)
    s_up, s_down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    kernels_done = [None] * NS                                      # per device slot: its batch's kernels have finished (dev_in free again)
    down_done = [None] * NS                                         # ... its download has finished (dev_out free again)
    state, index, k, out_off, pend = None, 0, 0, 0, None
    t_get = t_enq = t_slot = t_rel = 0.0
    marks = []
    try:
        while True:
            tt = time.perf_counter()
            item = reader.get()
            t_get += time.perf_counter() - tt
            if item is None:
                break
            i, n, got = item                                            # a trailing partial frame is dropped, as ffmpeg's rawvideo demuxer does
            tt = time.perf_counter()
            if n:
                d = k % NS
                # upload k on its own stream, once the kernels that last read this device slot (batch k - NS) are done
                if kernels_done[d] is not None:
                    s_up.wait_event(kernels_done[d])
                tim = a.staging_report
                if tim:
                    u0 = torch.cuda.Event(enable_timing=True); u0.record(s_up)
                reader.upload(i, n, dev_in[d], s_up)              # from the pinned slot, or straight from the registered file mapping
                up = torch.cuda.Event(enable_timing=tim)
                up.record(s_up)
                # kernels k behind the upload, and behind the download that last read this output slot
                compute.wait_event(up)
                if down_done[d] is not None:
                    compute.wait_event(down_done[d])
                if tim:
                    k0 = torch.cuda.Event(enable_timing=True); k0.record(compute)
                _, state = pipe.run(dev_in[d][:n], first_index=index, state=state, out=dev_out[d][:n])
                kd = torch.cuda.Event(enable_timing=tim)
                kd.record(compute)
                kernels_done[d] = kd
                # download k on the third stream into a free pinned slot; the writer thread takes it from there
                t_enq += time.perf_counter() - tt
                tt = time.perf_counter()
                j = writer.slot()
                t_slot += time.perf_counter() - tt
                tt = time.perf_counter()
                s_down.wait_event(kd)
                if tim:
                    d0 = torch.cuda.Event(enable_timing=True); d0.record(s_down)
                writer.download(j, n, dev_out[d], s_down)         # into the pinned slot, or straight into the registered output mapping
                dn = torch.cuda.Event(enable_timing=tim)
                dn.record(s_down)
                down_done[d] = dn
                if tim:
                    marks.append((u0, up, k0, kd, d0, dn, n))
                writer.put(j, n, dn, out_off if out_pos else None)
                out_off += n * frame_bytes
            # a pinned input slot goes back to the reader once its upload has completed: the PREVIOUS batch's is waited for here (long done),
            # so this thread never sits on the upload it has just enqueued
            t_enq += time.perf_counter() - tt
            tt = time.perf_counter()
            if pend is not None:
                pend[0].synchronize()
                reader.release(pend[1])
                pend = None
            t_rel += time.perf_counter() - tt
            if n:
                pend = (up, i)
            else:
                reader.release(i)
            index += n
            k += 1
            if got < B * frame_bytes:
                break
        if pend is not None:
            pend[0].synchronize()
            reader.release(pend[1])
        writer.close()
    except BaseException:
        # the render died (a read / write / HIP error, KeyboardInterrupt): everything queued on the GPU is drained, the writer is stopped, and a
        # regular output file is cut back to the frames that were actually written — in order, so a prefix — instead of keeping its planned length
        try:
            torch.cuda.synchronize(dev)
        except Exception:       # noqa: BLE001
            pass
        try:
            writer.close()
        except BaseException:   # noqa: BLE001 - the first error is the one to report
            pass
        try:
            reader.close()
        except BaseException:   # noqa: BLE001
            pass
        if out_pos:
            try:
                fout.flush()
                os.ftruncate(fout.fileno(), min(int(writer.frames), index) * frame_bytes)
            except OSError:
                pass
        raise
    t_pipe = time.perf_counter() - t_pipe
    reader.close()
    fout.flush()
    if out_pos:
        os.ftruncate(fout.fileno(), index * frame_bytes)      # an input that ended before its planned length, an output file that was longer
    if fout is not sys.stdout.buffer:
        fout.close()
    if fin is not sys.stdin.buffer:
        fin.close()
    print(f"{index} frames, elapsed {time.perf_counter() - t0:.3f}s", file=sys.stderr)      # ref:1269
    if a.staging_report and len(marks) > 4:
        torch.cuda.synchronize()
        mk = marks[2:]                   # past the start-up batches
        up_ms = sum(m[0].elapsed_time(m[1]) for m in mk) / len(mk)
        k_ms = sum(m[2].elapsed_time(m[3]) for m in mk) / len(mk)
        dn_ms = sum(m[4].elapsed_time(m[5]) for m in mk) / len(mk)
        span = mk[0][0].elapsed_time(mk[-1][5]) / len(mk)
        nb = sum(m[6] for m in mk) / len(mk) * frame_bytes
        print(f"staging (GPU side, per batch of {nb / frame_bytes:.0f} frames): upload {up_ms:.2f} ms = {nb / up_ms / 1e6:.1f} GB/s, kernels {k_ms:.2f} ms, "
              f"download {dn_ms:.2f} ms = {nb / dn_ms / 1e6:.1f} GB/s; one batch every {span:.2f} ms = {nb / frame_bytes / span * 1e3:.0f} frames/s", file=sys.stderr)
    if a.staging_report:
        # start-up (imports, ctx, tables, pinning the staging slots) and the exit are outside this figure; the reader's first read, the unoverlapped
        # legs of the first and last batch and the output's last write are inside it
        print(f"staging: pipeline {index} frames in {t_pipe:.3f} s = {index / max(t_pipe, 1e-9):.0f} frames/s (first read issued ... last batch written)", file=sys.stderr)
        print(f"staging: start-up imports {t_imports:.3f} s, device + ctx + tables {t_engine:.3f} s, files + staging slots {t_slots:.3f} s", file=sys.stderr)
        print(f"staging: input {reader.mapped_batches} batches mapped (hipHostRegister {reader.rmap.t_reg if reader.rmap else 0.0:.3f}s), {reader.staged_batches} staged"
              f"{' — ' + reader.note if reader.note else ''} | output {writer.mapped_batches} batches mapped (register + page allocation {writer.t_prep:.3f}s)"
              f"{' — ' + writer.note if writer.note else ''} | pipe buffers in {pipe_in} out {pipe_out} bytes", file=sys.stderr)
        print(f"staging: reader read {reader.t_io:.3f}s waited-for-slot {reader.t_wait:.3f}s | feeder waited-for-input {t_get:.3f}s enqueued {t_enq:.3f}s "
              f"waited-for-output-slot {t_slot:.3f}s waited-for-upload {t_rel:.3f}s | writer waited-for-download {writer.t_wait:.3f}s wrote {writer.t_io:.3f}s",
              file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
