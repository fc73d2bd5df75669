#!/usr/bin/env python3
"""bench.py — frames/sec of the per-frame CRT effect chain on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4}] [--batch B]

One "step" = one pass of the hot path (k_phosphor -> k_warp per frame, enqueued back to back by
crtfx_process_batch) over one batch of B synthetic frames already resident in HBM.  The default
workload is BASELINE.json configs[2] — the 4K full chain (Gaussian bloom sigma=3, warp 0.15), the
configuration the metric is quoted on.  N > 1: one process per GPU (torch.distributed.run), the
frame batches are sharded with no data-path collective for persistence 0 (weak scaling: every
rank processes its own B frames per step); --config 4 adds the one-frame persistence carry over
RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def synth_frames(b, h, w, device, seed=1234):
    """50 % smooth moving gradient + 50 % uniform noise, seed 1234+i, generated on device."""
    out = torch.empty((b, h, w, 3), dtype=torch.uint8, device=device)
    yy = torch.arange(h, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(w, device=device, dtype=torch.float32)[None, :]
    for i in range(b):
        g = torch.Generator(device=device).manual_seed(seed + i)
        noise = torch.randint(0, 256, (h, w, 3), device=device, dtype=torch.int32, generator=g)
        grad = torch.stack([((xx + 7 * i) % w) * (255.0 / w) + 0 * yy, ((yy + 5 * i) % h) * (255.0 / h) + 0 * xx,
                            ((xx + yy + 3 * i) % (h + w)) * (255.0 / (h + w))], dim=2).to(torch.int32)
        out[i] = ((noise + grad) // 2).clamp_(0, 255).to(torch.uint8)
    return out


def cpu_baseline(rs, h, w, fps, n_frames, seed=1234):
    """The oracle (CPU restatement of the reference chain: numpy + C for the OpenCV ops) on a
    bounded sample of the same workload, single thread.  A reported baseline, not the target."""
    from oracle import crt_oracle as orc
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    frames = []
    for i in range(n_frames):
        noise = rng.integers(0, 256, (h, w, 3), dtype=np.int32)
        grad = np.stack([((xx + 7 * i) % w) * (255.0 / w), ((yy + 5 * i) % h) * (255.0 / h), ((xx + yy + 3 * i) % (h + w)) * (255.0 / (h + w))], axis=2).astype(np.int32)
        frames.append(np.clip((noise + grad) // 2, 0, 255).astype(np.uint8))
    planes = [rng.standard_normal((h, w), dtype=np.float32) for _ in range(n_frames)] if rs.noise_strength > 0 else None
    params = dict(scanline_strength=rs.scanline_strength, triad_gamma=rs.triad_gamma, triad_preserve_luma=rs.triad_preserve_luma,
                  aberration_px=rs.aberration_px, bloom_sigma=rs.bloom_sigma, bloom_strength=rs.bloom_strength,
                  bloom_threshold=rs.bloom_threshold, noise_strength=rs.noise_strength, scanline_period_px=rs.scanline_period_px,
                  fast_bloom=rs.fast_bloom, pixel_size=rs.pixel_size, warp_strength=rs.warp_strength)
    orc._lib()
    t0 = time.perf_counter()
    orc.process_frames(frames, params, fps, rs.scanline_speed_px_s, rs.persistence, rs.triad_strength, rs.triad_softness,
                       rs.vignette_strength, noise_planes=planes)
    dt = time.perf_counter() - t0
    # the reference's own topology (ref:1015-1017, 1044-1131): static effects of the frames on a 2-thread pool (numpy and
    # the C restatement release the GIL), in-order persistence blend + quantise on the main thread
    from concurrent.futures import ThreadPoolExecutor
    triad = orc.make_triad_mask(h, w, rs.triad_strength, rs.triad_softness) if rs.triad_strength > 0.0 else None
    vig = orc.make_vignette(h, w, rs.vignette_strength) if rs.vignette_strength > 0.0 else None

    def static(i):
        return orc.apply_static_effects(frames[i], params["scanline_strength"], triad, float(params["triad_gamma"]), bool(params["triad_preserve_luma"]),
                                        params["aberration_px"], params["bloom_sigma"], params["bloom_strength"], float(params["bloom_threshold"]),
                                        params["noise_strength"], vig, params["scanline_period_px"], (i / float(fps)) * rs.scanline_speed_px_s,
                                        params["fast_bloom"], params["pixel_size"], 0, 0.0, time_sec=i / float(fps),
                                        warp_strength=float(params["warp_strength"]), noise_plane=None if planes is None else planes[i])
    t0 = time.perf_counter()
    prev = None
    with ThreadPoolExecutor(max_workers=2) as ex:
        for st in ex.map(static, range(n_frames)):
            prev, _ = orc.persistence_blend(prev, st, rs.persistence)
    dt2 = time.perf_counter() - t0
    return n_frames / dt, dt, n_frames / dt2


def copy_ceiling(device):
    """GB/s (read + write) of a 1 GiB device-to-device copy: the practical HBM ceiling on this box."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    return round(5 * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


MALL_COPY_EXE = os.path.join(ROOT, "build", "ubench", "mall_copy")      # tools/ubench/mall_copy.hip, compiled by __graft_entry__.build()


def _mall_rows(text, want):
    import re
    rows = {}
    for line in text.splitlines():
        m = re.match(r"(\d+) frame\(s\).*?\b(x4|x3|tap4|row2)\s+([\d.]+) us\s+(\d+) GB/s", line)
        if m and int(m.group(1)) == want:
            rows[m.group(2)] = {"us_per_group": float(m.group(3)), "gbs": float(m.group(4))}
    return rows


def mall_ceiling(frames_per_launch, h, w):
    """GB/s (float32 source + uint8 output bytes) at which a hand-written stream reads a launch group's pre-warp images back RIGHT AFTER they
    were written ('x3' = one 12-byte pixel per lane; 'tap4' = k_warp_lean's four-tap access shape with no arithmetic): k_warp's yardstick — its
    source is kept under the 256 MB Infinity Cache on purpose, so the HBM-sized copy of `copy_ceiling` is the wrong one for it.  MEASURED ON THIS
    BOX, before the timed region: tools/ubench/mall_copy.hip (built in-tree by build()) run as a child process for this frame size and group
    (~1 s).  None when the program is not built or fails — the committed figure of another box is then only under `reference_figures`."""
    import subprocess
    want = max(1, int(round(frames_per_launch)))
    if not os.access(MALL_COPY_EXE, os.X_OK):
        return None
    try:
        r = subprocess.run([MALL_COPY_EXE, str(int(w)), str(int(h)), str(want)], capture_output=True, text=True, timeout=60)
    except (OSError, subprocess.TimeoutExpired):
        return None
    rows = _mall_rows(r.stdout, want) if r.returncode == 0 else {}
    if "x3" not in rows:
        return None
    return {"gbs": rows["x3"]["gbs"], "us_per_group": rows["x3"]["us_per_group"], "frames": want,
            "tap4_us_per_group": rows.get("tap4", {}).get("us_per_group"),
            "source": "tools/ubench/mall_copy.hip run on this box before the timed region"}


def mall_ceiling_committed(frames_per_launch, h):
    """The same figure from a committed run on ANOTHER box of the same chip model (profiles/r04_mall_copy*.txt): a reference figure, kept out
    of `roofline`."""
    name = {2160: "r04_mall_copy.txt", 1080: "r04_mall_copy_1080p.txt"}.get(int(h))
    if name is None:
        return None
    want = max(1, int(round(frames_per_launch)))
    try:
        rows = _mall_rows(open(os.path.join(ROOT, "profiles", name)).read(), want)
    except OSError:
        return None
    if "x3" not in rows:
        return None
    return {"gbs": rows["x3"]["gbs"], "us_per_group": rows["x3"]["us_per_group"], "frames": want, "source": f"profiles/{name} (committed run, another box)"}


def preflight_need_bytes(h, w, B, elem_bytes, persistence, out_slots, keep_states):
    """Device memory one rank of the bench allocates: the B resident input frames, the engine's output slots, the float32 local states it
    keeps per slot (+ chunk-final / zero / carry frames), the library's pre-warp scratch (<= 2 x 224 MB) and the 2 GiB of `copy_ceiling`."""
    frame = h * w * 3
    need = B * frame * elem_bytes                      # frames resident for the whole run
    need += B * frame if elem_bytes > 1 else 0         # half frames are synthesised as uint8 and converted: both copies exist for a moment
    need += 8 * frame * 16                             # synth_frames' per-frame int32 / float32 temporaries
    need += out_slots * B * frame * elem_bytes
    if persistence > 0.0:
        need += (out_slots * (keep_states + 1) + 3) * frame * 4
    need += 2 * (224 << 20) + (2 << 30)
    return need


def preflight(a, world, rank, local_rank, backend, h, w, B, elem_bytes, persistence, out_slots, keep_states):
    """Checks made by every rank BEFORE the rendezvous and before anything large is allocated; returns a one-line reason or None.  A rank
    that fails exits non-zero at once, and torch.distributed.run then tears the other ranks down: a mis-sized or mis-launched N-rank run ends
    in seconds with a reason instead of hanging at the rendezvous or dying in an allocation halfway through the warm-up."""
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        return f"--gpus {world} over RCCL needs {world} visible devices, this rank sees {ndev} (CRTFX_DIST_BACKEND=gloo rehearses more ranks than GPUs)"
    if backend == "nccl" and local_rank >= ndev:
        return f"LOCAL_RANK {local_rank} has no device ({ndev} visible)"
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ndev)
    free, total = torch.cuda.mem_get_info(dev_index)
    sharing = 1 if backend == "nccl" else -(-world // max(1, ndev))       # gloo rehearsal: several ranks share one device
    need = preflight_need_bytes(h, w, B, elem_bytes, persistence, out_slots, keep_states)
    if need * sharing > free:
        return (f"rank {rank}: {B} frames of {w}x{h} per step need {need / 2**30:.1f} GiB of device memory"
                f"{' x ' + str(sharing) + ' ranks on this device' if sharing > 1 else ''}, {free / 2**30:.1f} of {total / 2**30:.1f} GiB are free — lower --batch")
    return None


class GpuTelemetry:
    """Shader clock and package power of THIS rank's GPU, sampled from a host thread while the timed region runs (librocm_smi64 through
    ctypes — nothing on the GPU stream, no subprocess).  Why it is in the line: with eight packages drawing ~1.3 kW each at full load
     a node-level power or thermal cap is the one plausible way for the frame-sharded path to come in under 7x at
    8 GPUs, and per-rank frames/s alone could not tell that from a hop stall or a slow host.  Every failure degrades to nulls."""

    def __init__(self, device, period_s=0.05):
        import ctypes as C
        import threading
        self.err, self.samples, self._stop, self._thr = None, [], threading.Event(), None
        self.period = period_s
        try:
            self.lib = C.CDLL("librocm_smi64.so")
            if self.lib.rsmi_init(C.c_uint64(0)) != 0:
                raise OSError("rsmi_init failed")
            n = C.c_uint32(0)
            self.lib.rsmi_num_monitor_devices(C.byref(n))
            pr = torch.cuda.get_device_properties(device)
            want = (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", 0))
            self.idx = None
            for i in range(n.value):
                bdf = C.c_uint64(0)
                if self.lib.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(bdf)) == 0:
                    v = bdf.value
                    if ((v >> 32) & 0xFFFFFFFF, (v >> 8) & 0xFF, (v >> 3) & 0x1F) == want:
                        self.idx = i
                        break
            if self.idx is None:
                self.idx = device.index if n.value > (device.index or 0) else 0
                self.err = "PCI id not matched: device index used"

            class Freqs(C.Structure):
                _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]
            self._Freqs, self._C = Freqs, C
        except Exception as e:       # noqa: BLE001 - telemetry must never take the bench down
            self.lib, self.err = None, f"{type(e).__name__}: {e}"

    def _sample(self):
        C = self._C
        f, pw = self._Freqs(), C.c_uint64(0)
        mhz = watts = None
        if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.idx), C.c_int(0), C.byref(f)) == 0 and f.current < 33:
            mhz = f.frequency[f.current] / 1e6
        if self.lib.rsmi_dev_current_socket_power_get(C.c_uint32(self.idx), C.byref(pw)) == 0:
            watts = pw.value / 1e6
        return mhz, watts

    def start(self):
        import threading
        if self.lib is None:
            return
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                try:
                    self.samples.append(self._sample())
                except Exception as e:       # noqa: BLE001
                    self.err = f"{type(e).__name__}: {e}"
                    return
                self._stop.wait(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=5)
            self._thr = None
        clk = [m for m, _ in self.samples if m]
        pw = [w_ for _, w_ in self.samples if w_]
        return {"samples": len(self.samples), "period_s": self.period,
                "sclk_mhz_mean": round(sum(clk) / len(clk), 1) if clk else None, "sclk_mhz_min": round(min(clk), 1) if clk else None,
                "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None,
                **({"note": self.err} if self.err else {})}


def source_hash():
    """Fingerprint of the device code a PMC measurement belongs to (profiles/traffic.json carries the same field):
    sha1 over the kernel sources, so that a stale traffic figure is never attached to a different build."""
    import hashlib
    h = hashlib.sha1()
    for f in ("pythoncrt_amd/csrc/crtfx_kernels.hip.h", "pythoncrt_amd/csrc/crtfx_common.hip.h", "pythoncrt_amd/csrc/crtfx_blur.hip.h",
              "pythoncrt_amd/csrc/crtfx_point.hip.h", "pythoncrt_amd/csrc/crtfx_phosphor.hip.h", "pythoncrt_amd/csrc/crtfx_phosphor_ct.hip.h", "pythoncrt_amd/csrc/crtfx_warp.hip.h",
              "pythoncrt_amd/csrc/crtfx.hip", "pythoncrt_amd/csrc/crtfx_rr.hip", "pythoncrt_amd/csrc/crtfx_internal.h", "include/crtfx.h"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def backend_version(backend):
    """Version string of the collective library the ranks talk through (RCCL reports itself through torch's nccl binding)."""
    try:
        if backend == "nccl":
            return "rccl " + ".".join(str(x) for x in torch.cuda.nccl.version())
        return f"gloo (torch {torch.__version__})"
    except Exception as e:      # noqa: BLE001 - a version string must never cost a bench line
        return f"unknown ({e.__class__.__name__})"


class stdout_to_stderr:
    """File-descriptor-level redirect of stdout to stderr for the duration of the block.  RCCL 2.26 prints a five-line banner ("RCCL version :
    ...", HIP / ROCm versions, host name, library path) on STDOUT from rank 0 when its first communicator comes up — measured on the one-GPU
    box in round 6 (profiles/r06_rccl_world1.json's run) — and the contract of this program is ONE JSON line on stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=[0, 2, 3, 4, 5], help="BASELINE.json configs[N-1]; 0 = the reference CLI's default flags at 1080p")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default: 1920 at 4K, 8192 at 1080p, 384 at 8K — sized so that the default 20 steps run >= 3 s)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames in the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--repeats", type=int, default=2, help="further timed regions of K steps after the reported one (spread only; 0 = none)")
    ap.add_argument("--tables-outside", action="store_true", help="build the per-frame host tables before the timed region (A/B of the host share)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py launches the ranks itself")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: take the N > 1 code path at world size 1 — init_process_group over RCCL, the barrier / all_reduce / "
                         "all_gather calls, and (config 4) the sharded-persistence schedule as the one-rank ring, its hop a real isend / irecv to itself")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="libcrtfx testing/tuning switch (crtfx_set_option), e.g. NO_CC=1; never part of a reported result")
    return ap.parse_args(argv)


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes
    (python -m torch.distributed.run ... bench.py <same flags>) and relay rank 0's JSON line.  This parent never
    touches the GPU (no torch.cuda / HIP call), so nothing that has initialised a device is ever replaced or forked."""
    import socket
    import subprocess
    port = a.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [x for x in sys.argv[1:]]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)
    print(line)
    raise SystemExit(0)


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                      # never returns
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.force_dist and world != 1:
        raise SystemExit("--force-dist is the world-size-1 rehearsal of the N > 1 path (use --gpus N for N > 1)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU fallback)")
    use_dist = world > 1 or a.force_dist
    backend = os.environ.get("CRTFX_DIST_BACKEND", "nccl")      # "gloo": rehearsal of several ranks on fewer GPUs (never a result)

    from pythoncrt_amd import effects
    from pythoncrt_amd.pipeline import FramePipeline, GpuShardEngine, baseline_config
    from pythoncrt_amd.shard import FrameShard, ShardedRender, settle_frames
    for o in a.opt:
        k, v = o.split("=", 1)
        effects.DEBUG_OPTIONS[k.strip().upper()] = int(v)
    rs, h, w = baseline_config(a.config)
    fps = 30.0
    p = rs.persistence
    # ---- pre-flight: device count and device memory, before the rendezvous (a failing rank exits at once with a one-line reason) ----
    B_plan = a.batch or (384 if h >= 4320 else 1920 if h >= 2160 else 8192)
    if p > 0.0 and not a.batch:
        B_plan = max(settle_frames(p), 4096 if world > 1 else 8192)
    why = preflight(a, world, rank, local_rank, backend, h, w, B_plan, 2 if a.config == 5 else 1, p, 2 if (p > 0.0 and use_dist) else 1,
                    min(B_plan, settle_frames(p, 2.0 ** -26)) if p > 0.0 else 0)
    if why:
        print(f"bench.py preflight failed: {why}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    device = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist = None
    if use_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # --force-dist with no launcher around it: a free port of our own
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        # a rank that never arrives (it failed its pre-flight, or was never started) must not leave the others at the rendezvous for the
        # default half hour: one minute, then init_process_group raises and the job ends non-zero
        with stdout_to_stderr():     # the communicator comes up here (device_id: eager) or at the first collective: both inside the redirect
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("CRTFX_RENDEZVOUS_TIMEOUT_S", "60"))),
                                    **({"device_id": device} if backend == "nccl" else {}))
            dist.barrier()
    # frames per step: enough that the default 20 steps run >= 3 s — a timed region the driver's 5-s GPU-busy sampler
    # cannot miss (4K: 1920 frames = 47.8 GB in + 47.8 GB out of the 288 GB; 1080p: 8192 frames = 51 GB each way; 8K fp16:
    # 384 frames = 76.4 GB each way).  Sharded persistence needs B >= settle_frames(p) anyway (shard.py); per-frame states
    # are kept for the first 26 frames of a chunk only (GpuShardEngine.keep), so a long chunk costs output frames, not states.
    B = a.batch or (384 if h >= 4320 else 1920 if h >= 2160 else 8192)
    if p > 0.0 and not a.batch:
        B = max(settle_frames(p), 4096 if world > 1 else 8192)
    dtype = torch.float16 if a.config == 5 else torch.uint8
    pipe = FramePipeline(device, h, w, rs, fps=fps, noise_seed=1234, dtype=dtype)
    frames = synth_frames(B, h, w, device, seed=1234 + 1000 * rank).to(dtype)
    shard = FrameShard(world, rank, B)
    sharded_iir = p > 0.0 and use_dist
    engine = GpuShardEngine(pipe, B, slots=2 if sharded_iir else 1)
    # the overlapped hop schedule is the default over gloo (rehearsals) only: over RCCL it has run at world size 1 (the one-rank ring,
    # tests/test_rccl_world1_gpu.py) but never between two GPUs, and at these chunk sizes the synchronous hop costs < 2 % of a round
    # (0.16 ms of one xGMI link + 0.15 ms of fix-up against a >= 20 ms scan; world 1: 0.14 %); CRTFX_SHARD_OVERLAP=1 opts in.
    want_overlap = sharded_iir and (backend == "gloo" or os.environ.get("CRTFX_SHARD_OVERLAP") == "1")
    render = ShardedRender(shard, p, engine, dist=dist, overlap=want_overlap, timing=sharded_iir, loopback=bool(a.force_dist))

    # step s = round s of the frame-sharded render: rank r owns global frames [(s*world + r)*B, ... + B).
    # The per-frame host tables of a step (scanline row gains via np.sin, flicker factors, the ctypes frame records and
    # their upload) are built INSIDE the timed region, as a render has to; the GPU works on step s meanwhile.
    n_regions = 1 + max(0, a.repeats)
    total_steps = a.warmup + a.steps * n_regions
    host_s = [0.0]

    def build_tables(s_):
        t = time.perf_counter()
        engine.records[(s_ * world + rank) * B] = pipe.frame_records((s_ * world + rank) * B, B)
        host_s[0] += time.perf_counter() - t

    if a.tables_outside:
        for s_ in range(total_steps):
            build_tables(s_)

    def one_step(s_):
        if not a.tables_outside:
            build_tables(s_)
        render.submit_round(frames, s_)        # overlapped schedule when asked for (results one call late), else run_round

    def sync():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for s in range(a.warmup):
        one_step(s)
    render.flush()
    sync()
    # k_warp's yardstick, measured on this box between the warm-up and the timed region (rank 0; a child process, ~1 s)
    mall_now = None
    if rank == 0 and rs.warp_strength != 0.0 and not a.no_profile:
        mall_now = mall_ceiling(pipe.plan().get("group_max", 1), h, w)
    if dist is not None:
        dist.barrier()
    prof = not a.no_profile
    region_dt = []
    kt = {}
    own_dt = None
    telem = GpuTelemetry(device)
    telem_first = None
    for reg in range(n_regions):
        first = a.warmup + reg * a.steps
        host_s[0] = 0.0
        if reg == 0:
            pipe.profile(8 if prof else 0)      # HIP events on the launches of every 8th frame of the reported region
            telem.start()
        t0 = time.perf_counter()
        for s in range(first, first + a.steps):
            one_step(s)
        render.flush()
        sync()
        dt_r = time.perf_counter() - t0
        if reg == 0:
            telem_first = telem.stop()
            kt = pipe.profile_read() if prof else {}
            pipe.profile(False)
            host_first = host_s[0]
        if reg == 0:
            own_dt = dt_r                       # this rank's own wall time of the reported region (before the max over ranks)
        if dist is not None:
            t = torch.tensor([dt_r], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        region_dt.append(dt_r)
    dt = region_dt[0]

    # what every rank saw of the reported region — frames/s on its own clock, its GPU's shader clock and package power, its kernel
    # time per launch group and (sharded persistence) its own hop / fix-up split: a sub-7x result at 8 GPUs can then be attributed from
    # the line alone (throttling: sclk / power differ between ranks or from the 1-GPU line; hop stall: shard_schedule; host: frames/s
    # below what chain_ms_per_launch_group allows)
    kt_ok = {k: v for k, v in (kt or {}).items() if v[1] and v[2]}
    mine_rec = {"rank": rank, "frames_per_s": round(B * a.steps / region_dt[0], 2) if region_dt else None,
                "frames_per_s_own_clock": round(B * a.steps / own_dt, 2) if own_dt else None,
                "chain_ms_per_frame": round(sum(v[0] * v[1] / v[2] for v in kt_ok.values()), 5) if kt_ok else None,
                "gpu": telem_first,
                **({"shard_schedule": render.schedule_report()} if sharded_iir else {})}
    per_rank = None
    per_rank_detail = [mine_rec]
    if dist is not None:
        mine = torch.tensor([B * a.steps / dt], dtype=torch.float64, device=device)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [round(float(v.item()), 2) for v in allv]
        per_rank_detail = [None] * world
        dist.all_gather_object(per_rank_detail, mine_rec)

    total_frames = B * a.steps * world
    fps_out = total_frames / dt
    px = h * w
    survey_bytes_frame = px * ((12 if a.config == 5 else 6) + (24 if p > 0 else 0))   # u8 (fp16) in + out (+ f32 state r/w per frame)
    # ... but the kernels carry the persistence state in registers through a run of up to CRTFX_MAX_GROUP = 8 frames
    # (DESIGN.md section 3): the float32 state is read once and written once per RUN, so the bytes the chain has to move
    # per frame are in + out + 24 B/px / 8.  `roofline.frac` is quoted on that figure; the survey's 30 B/px stays beside it.
    run_len = 8
    alg_bytes_frame = px * (12 if a.config == 5 else 6) + (px * 24 // run_len if p > 0 else 0)
    res = {
        "metric": "4K frames/sec (whole node) + achieved HBM GB/s as % of MI355X peak" if a.config == 3 else f"{h}p frames/sec (whole node)",
        "value": round(fps_out, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",      # the arithmetic type of the chain up to the vignette; config.arithmetic says what runs after it
        "config": {"workload": (f"BASELINE configs[{a.config - 1}]" if a.config else "reference CLI defaults (fast bloom, pixel_size 2)") + f": {w}x{h} chain (scanlines+triad+aberration+bloom sigma={rs.bloom_sigma}"
                               f"+warp {rs.warp_strength}+vignette+grain), persistence {p}, {'fp16' if a.config == 5 else 'u8'} in/out",
                   "arithmetic": "float32 through the bloom, triad and scanline stages; float64 for the vignette / flicker / grain tail and the warp's "
                                 "bilinear sums (NumPy's promotion, ref:626-647), narrowed to float32 where a value is stored",
                   "frames_per_step_per_gpu": B, "parallelism": f"frame-shard x{world}"},
        "timed_region_s": round(dt, 4),
        **({"tuning_options": dict(effects.DEBUG_OPTIONS)} if effects.DEBUG_OPTIONS else {}),
        "host_tables": {"in_timed_region": not a.tables_outside, "host_seconds_rank0": round(host_first, 4),
                        "note": "per-frame scanline/flicker tables + frame records built and uploaded per step; they overlap the previous step's kernels"},
        "repeat_values": [round(total_frames / d, 2) for d in region_dt[1:]],
        "dist": {"backend": backend if use_dist else None, "backend_version": backend_version(backend) if use_dist else None,
                 **({"forced": True, "note": "--force-dist: the N > 1 code path at world size 1 (a rehearsal of the collectives and the hop, not a scaling result)"} if a.force_dist else {}),
                 "world_size_seen": (dist.get_world_size() if dist is not None else 1), "visible_devices": torch.cuda.device_count(),
                 # how the one-frame persistence hop ran (config 4 at N > 1 only): behind each round's scan ("synchronous") or beside the next one
                 "hop_schedule": (("overlapped" if render.overlap else "synchronous") if sharded_iir else None),
                 "per_rank_frames_per_s": per_rank, "per_rank": per_rank_detail},
    }
    if sharded_iir:
        res["shard_schedule"] = render.schedule_report()
    if rank == 0:
        if kt:
            # kt[k] = (mean ms per launch, timed launches, frames they covered): a launch of the dominant kernel
            # covers fpl frames (crtfx_process_batch groups frames per grid), so everything below is quoted PER LAUNCH
            # of that kernel = per group of fpl frames, with the chain's other kernels over the same frames added.
            kt = {k: v for k, v in kt.items() if v[1] and v[2]}
            per_frame_ms = {k: v[0] * v[1] / v[2] for k, v in kt.items()}
            dom = max(per_frame_ms, key=per_frame_ms.get)
            fpl = kt[dom][2] / kt[dom][1]
            group_ms = sum(per_frame_ms.values()) * fpl
            if p > 0 and "k_warp" in kt:
                run_len = max(1.0, kt["k_warp"][2] / kt["k_warp"][1])       # frames per persistence run = frames per k_warp launch
                alg_bytes_frame = px * (12 if a.config == 5 else 6) + int(px * 24 / run_len)
            alg_launch = alg_bytes_frame * fpl
            achieved = alg_launch / (group_ms * 1e-3) / 1e9
            src_now = source_hash()

            def evidence(name):
                """profiles/<name>.json entry of this config, only when it was measured on THIS device code."""
                path = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(path):
                    return None, None
                try:
                    ent = json.load(open(path)).get(f"config{a.config}")
                except Exception:
                    return None, None
                if not isinstance(ent, dict):
                    return None, None
                if ent.get("source_hash") != src_now:
                    return None, {"stale": True, "measured_on": ent.get("source_hash"), "this_build": src_now}
                return ent, {k: ent.get(k) for k in ("source_hash", "tag", "batch", "correction") if k in ent}

            # PMC traffic: FETCH_SIZE + WRITE_SIZE of separate --pmc passes.  These counters sit between L2 and the fabric: they
            # count Infinity-Cache (MALL) hits as well as HBM accesses (MI355X_MICROARCH.md, HBM / rocprofv3 section), so the rate
            # below is FABRIC bandwidth — an upper bound on the HBM traffic, which for a <= 224 MB launch group held under the
            # 256 MB Infinity Cache is probably close to the algorithmic bytes.
            tent, tsrc = evidence("traffic.json")
            traffic = int(tent["bytes_per_frame_corrected"] * fpl) if tent else None
            traffic_raw = int(tent["bytes_per_frame_as_reported"] * fpl) if tent else None
            # VALU / LDS pipe occupancy of the chain from the SQ counter passes (tools/summarise_profiles.py)
            vent, vsrc = evidence("valu.json")
            valu = lds = None
            bound, bound_ev = "hbm", {"note": "no SQ counter evidence for this build (profiles/valu.json missing or measured on other sources)"}
            if vent:
                clk = vent.get("clock_ghz", 2.4)
                chain_cycles = (group_ms / fpl) * 1e-3 * clk * 1e9              # shader cycles the chain spends per frame
                wi = vent["valu_wave_insts_per_frame"]
                cw = vent.get("valu_cost_weighted_cycles_per_frame")
                valu = {"wave_insts_per_frame": int(wi),
                        # literal issue fraction: every wave-instruction priced at 2 cycles (SIMD-32)
                        "issue_frac": round(wi * 2.0 / (1024 * chain_cycles), 4),
                        # instructions weighted by their measured issue cost (profiles/r02_valu_cost.txt x the static mix of each
                        # kernel's ISA, tools/isa_cost.py): the fraction of the chain's SIMD time the VALU is busy
                        "cost_weighted_frac": round(cw / (1024 * chain_cycles), 4) if cw else None,
                        "avg_cycles_per_inst": vent.get("valu_avg_cycles_per_inst"),
                        "clock_ghz": clk, "source": vsrc}
                la = vent.get("lds_idx_active_cycles_per_frame")
                if la:
                    lds = {"pipe_frac": round(la / (256 * chain_cycles), 4),        # SQ_LDS_IDX_ACTIVE over 256 CUs x chain cycles
                           "bank_conflict_share": round(vent.get("lds_bank_conflict_cycles_per_frame", 0) / la, 4),
                           "dominant_kernel_pipe_frac": vent.get("dominant_lds_pipe_frac"), "source": vsrc}
                cand = {"valu": (valu or {}).get("cost_weighted_frac") or 0.0, "lds": (lds or {}).get("pipe_frac") or 0.0,
                        "fabric": (traffic / (group_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else 0.0,
                        "hbm": achieved / HBM_PEAK_GBS}
                order = sorted(cand, key=cand.get, reverse=True)
                bound = order[0] if cand[order[0]] - cand[order[1]] > 0.15 else f"{order[0]}+{order[1]}"
                bound_ev = {"fractions_of_each_limit": {k: round(v, 4) for k, v in cand.items()},
                            "rule": "the busiest resource; two named when within 0.15 of each other",
                            "dominant_kernel_wave_time": vent.get("dominant_wave_time"), "source": vsrc}
            fabric = round(traffic / (group_ms * 1e-3) / 1e9, 1) if traffic else None
            res["roofline"] = {
                # what the counters say limits the chain (evidence beside it); achieved / peak / frac below stay the HBM
                # figures the metric asks for: ALGORITHMIC bytes over kernel time against 8 TB/s
                "bound": bound, "bound_evidence": bound_ev,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_as_reported": traffic_raw, "traffic_source": tsrc,
                "traffic_note": "FETCH_SIZE + WRITE_SIZE per launch group: L2 <-> Infinity Cache / HBM bytes (MALL hits included), not HBM alone",
                "kernel": f"{dom} (+ the other kernels of the chain over the same frames): algorithmic bytes of the frames of one launch / their summed kernel time",
                "frames_per_launch": round(fpl, 3), "algorithmic_bytes_per_launch": int(alg_launch),
                "algorithmic_bytes_per_frame": int(alg_bytes_frame),
                **({"survey_bytes_per_frame": int(survey_bytes_frame),
                    "persistence_note": "the float32 state is read and written once per run of frames_per_run frames (it stays in registers inside a run), "
                                        "not once per frame as SURVEY 8d budgets (30 B/px): frac is on the bytes the kernels have to move",
                    "frames_per_run": round(run_len, 3)} if p > 0 else {}),
                "chain_ms_per_launch_group": round(group_ms, 4),
                "dominant_kernel": dom,
                # the same algorithmic bytes over the dominant kernel's launch duration ALONE (the literal per-kernel reading;
                # `achieved` above also charges the other kernels of the chain and is the smaller, conservative figure)
                "dominant_kernel_only": {"avg_launch_ms": round(kt[dom][0], 4),
                                         "achieved": round(alg_launch / (kt[dom][0] * 1e-3) / 1e9, 1),
                                         "frac": round(alg_launch / (kt[dom][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                # SURVEY 8d's second figure, named for what the counters see: fabric bytes over the same kernel time, next to what a
                # plain device copy reaches on this box
                "fabric_achieved": fabric,
                "fabric_frac": round(fabric / HBM_PEAK_GBS, 4) if fabric else None,
                "fabric_frac_as_reported": round(traffic_raw / (group_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic_raw else None,
                "valu": valu, "lds": lds,
                "copy_ceiling": copy_ceiling(device),
                # ... and what reading one launch group's float32 pre-warp images back out of the Infinity Cache reaches (k_warp's yardstick)
                "mall_ceiling": mall_now,
                "kernels": {k: {"avg_launch_ms": round(v[0], 4), "timed_launches": v[1], "frames_per_launch": round(v[2] / v[1], 3)}
                            for k, v in kt.items()},
                "plan": pipe.plan(),      # crtfx_last_plan: the kernel builds the timed launches landed on
            }
            # figures NOT measured in this run (another box of the same chip model): context only, never part of `roofline`
            res["reference_figures"] = {"mall_ceiling_committed": mall_ceiling_committed(fpl, h) if rs.warp_strength != 0.0 else None}
        if world == 1 and a.cpu_frames != 0 and a.config != 5:
            n_cpu = a.cpu_frames if a.cpu_frames > 0 else (8 if h >= 2160 else 32)   # ~10-12 s of CPU work (single thread + the 2-worker repeat)
            v, secs, v2 = cpu_baseline(rs, h, w, fps, n_cpu)
            res["cpu_baseline"] = {"value": round(v, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
                                   "sample": f"{n_cpu} frames of the same {w}x{h} workload through oracle/ (numpy + C restatement of the OpenCV ops), "
                                             f"{secs:.1f} s on 1 of {os.cpu_count()} host cores",
                                   "reference_topology": {"workers": 2, "value": round(v2, 4),
                                                          "note": "the same sample on the reference's 2-thread pool + in-order blend (ref:1015-1017)"}}
        print(json.dumps(res))
    if dist is not None:
        render.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
