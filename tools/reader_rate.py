"""Where does the CLI's reader lose its rate?  cli._Reader over a tmpfs file of 4K frames: (a) slots released at once, (b) each batch uploaded
on a side stream and released when the upload has completed, (c) as (b) with a concurrent download stream busy.   python tools/reader_rate.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pythoncrt_amd import cli      # noqa: E402

h, w, B = 2160, 3840, 16
fb = h * w * 3
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 480
path = "/dev/shm/crtfx_reader_rate.rgb"
with open(path, "wb") as f:
    blk = np.random.default_rng(0).integers(0, 256, 8 * fb, dtype=np.uint8).tobytes()
    for _ in range(nfr // 8):
        f.write(blk)
dev = torch.device("cuda", 0)
d_in = [torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
d_out = torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev)
h_out = torch.empty((B, h, w, 3), dtype=torch.uint8).pin_memory()
s_up, s_down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run(mode, slots=3):
    fin = open(path, "rb", buffering=0)
    jobs = ((k * B * fb, B) for k in range(nfr // B))
    rd = cli._Reader(fin, True, jobs, (B, h, w, 3), fb, slots=slots)
    t0 = time.perf_counter()
    pend, k = None, 0
    while True:
        it = rd.get()
        if it is None:
            break
        i, n, got = it
        if mode == "a":
            rd.release(i)
            continue
        with torch.cuda.stream(s_up):
            d_in[k % 3][:n].copy_(rd.bufs[i][:n], non_blocking=True)
            up = torch.cuda.Event(); up.record(s_up)
        if mode == "c":
            with torch.cuda.stream(s_down):
                h_out.copy_(d_out, non_blocking=True)
        if pend is not None:
            pend[0].synchronize(); rd.release(pend[1])
        pend = (up, i)
        k += 1
    if pend is not None:
        pend[0].synchronize(); rd.release(pend[1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rd.close(); fin.close()
    print(f"mode {mode} slots {slots}: {nfr} frames in {dt:.3f} s = {nfr / dt:.0f} frames/s = {nfr * fb / dt / 1e9:.1f} GB/s; reader read {rd.t_io:.3f} s "
          f"({nfr * fb / rd.t_io / 1e9:.1f} GB/s while reading), waited for a slot {rd.t_wait:.3f} s")


for mode in ("a", "a", "b", "c"):
    run(mode)
run("b", slots=4)
os.environ["X"] = "1"
cli._IO_POOL = None
import concurrent.futures
cli._IO_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=16)
run("a"); run("b")
os.unlink(path)
