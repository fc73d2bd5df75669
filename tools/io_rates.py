"""How fast can the host move a raw rgb24 file out of the page cache (tmpfs) into a pinned staging buffer?  The FIRST read of a file that
was just written and later reads differ (shmem pages are activated on their first read), so every line writes a fresh file, reads it twice
and reports both; os.preadv on 1..32 threads at several slice sizes, and mmap + memcpy for comparison.
    python tools/io_rates.py [MB]"""
import concurrent.futures
import mmap
import os
import sys
import time

import numpy as np
import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
n = mb << 20
path = "/dev/shm/crtfx_io_rates.bin"
blk = np.random.default_rng(0).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
pinned_t = torch.empty(n, dtype=torch.uint8).pin_memory()
pinned = pinned_t.numpy()
pinned[:] = 1
POOLS = {}


def fresh():
    if os.path.exists(path):
        os.unlink(path)
    with open(path, "wb") as f:
        for _ in range(n // len(blk)):
            f.write(blk)


def pool(th):
    if th not in POOLS:
        POOLS[th] = concurrent.futures.ThreadPoolExecutor(th)
        list(POOLS[th].map(lambda i: time.sleep(0.01), range(th)))      # threads started before anything is timed
    return POOLS[th]


def read_all(fd, threads, slice_bytes):
    view = memoryview(pinned)

    def one(lo):
        hi, got = min(n, lo + slice_bytes), 0
        while lo + got < hi:
            k = os.preadv(fd, [view[lo + got:hi]], lo + got)
            if k <= 0:
                break
            got += k
    t = time.perf_counter()
    list(pool(threads).map(one, range(0, n, slice_bytes)))
    return n / (time.perf_counter() - t) / 1e9


def mmap_copy(fd, threads, slice_bytes):
    m = mmap.mmap(fd, n, prot=mmap.PROT_READ)
    src = np.frombuffer(m, dtype=np.uint8)

    def one(lo):
        hi = min(n, lo + slice_bytes)
        np.copyto(pinned[lo:hi], src[lo:hi])
    t = time.perf_counter()
    list(pool(threads).map(one, range(0, n, slice_bytes)))
    r = n / (time.perf_counter() - t) / 1e9
    del src
    m.close()
    return r


print(f"{mb} MiB file in tmpfs -> pinned buffer; os.cpu_count() = {os.cpu_count()}; first read / second read of a freshly written file, GB/s")
for sl in (4 << 20, 16 << 20):
    for th in (1, 4, 8, 16, 32):
        fresh()
        fd = os.open(path, os.O_RDONLY)
        a, b = read_all(fd, th, sl), read_all(fd, th, sl)
        os.close(fd)
        print(f"preadv slice {sl >> 20:2d} MiB {th:2d} threads: {a:5.1f} / {b:5.1f}")
for th in (8, 16):
    fresh()
    fd = os.open(path, os.O_RDONLY)
    a, b = mmap_copy(fd, th, 16 << 20), mmap_copy(fd, th, 16 << 20)
    os.close(fd)
    print(f"mmap + numpy copy, 16 MiB slices, {th:2d} threads: {a:5.1f} / {b:5.1f}")
os.unlink(path)
