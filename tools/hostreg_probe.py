"""Dev tool (GPU box): can the raw-rgb24 edge DMA straight out of / into the page cache?

hipHostRegister on a MAP_SHARED mapping of a file (tmpfs and the box's /tmp), for the input (PROT_READ) and the output (ftruncate + PROT_WRITE):
does it register, how long does the registration take per byte, and what does hipMemcpyAsync reach on it — against the pinned-slot staging the
CLI uses today (page cache -> pinned by memcpy on the I/O threads -> DMA).     python tools/hostreg_probe.py [MiB]  ->  stdout"""
import ctypes
import mmap
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    return ctypes.CDLL("libamdhip64.so")


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1592          # 64 4K frames
    nbytes = mib << 20
    batch = 3840 * 2160 * 3 * 16
    dev = torch.device("cuda", 0)
    dbuf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    lib = hip()
    lib.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    lib.hipHostUnregister.argtypes = [ctypes.c_void_p]
    lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.hipGetErrorString.restype = ctypes.c_char_p
    lib.hipGetErrorString.argtypes = [ctypes.c_int]
    lib.hipGetLastError.restype = ctypes.c_int
    H2D, D2H = 1, 2
    stream = torch.cuda.current_stream(dev).cuda_stream
    print(f"{mib} MiB; os.sched_getaffinity: {len(os.sched_getaffinity(0))} cpus; page size {mmap.PAGESIZE}")

    def copy_rate(host_ptr, kind, label):
        torch.cuda.synchronize()
        for rep in range(2):
            t = time.perf_counter()
            for off in range(0, nbytes, batch):
                n = min(batch, nbytes - off)
                a, b = (dbuf.data_ptr() + off, host_ptr + off) if kind == H2D else (host_ptr + off, dbuf.data_ptr() + off)
                rc = lib.hipMemcpyAsync(a, b, n, kind, stream)
                if rc:
                    print(f"    {label}: hipMemcpyAsync -> {rc} {lib.hipGetErrorString(rc).decode()}")
                    lib.hipGetLastError()
                    return
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print(f"    {label} pass {rep}: {nbytes / dt / 1e9:.1f} GB/s")

    # baseline: torch pinned memory
    pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    print("pinned (hipHostMalloc):")
    copy_rate(pin.data_ptr(), H2D, "H2D")
    copy_rate(pin.data_ptr(), D2H, "D2H")
    del pin

    rng = np.random.default_rng(0)
    blk = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    for d in ("/dev/shm", "/tmp", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")):
        if not os.path.isdir(d):
            continue
        path = os.path.join(d, "crtfx_hostreg_probe.bin")
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < 2.2 * nbytes:
                print(f"{d}: only {st.f_bavail * st.f_frsize >> 20} MiB free, skipped")
                continue
            fs = next((l.split()[2] for l in reversed(open("/proc/mounts").read().splitlines()) if d.startswith(l.split()[1])), "?")
            with open(path, "wb") as f:
                for _ in range(mib):
                    f.write(blk)
            print(f"{d} ({fs}):")
            # ---- input: read-only shared mapping
            fd = os.open(path, os.O_RDONLY)
            m = mmap.mmap(fd, nbytes, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
            arr = np.frombuffer(m, dtype=np.uint8)
            ptr = arr.ctypes.data
            for flags, name in ((0, "default"), (8, "ReadOnly"), (1, "Portable")):
                t = time.perf_counter()
                rc = lib.hipHostRegister(ptr, nbytes, flags)
                dt = time.perf_counter() - t
                print(f"  input  PROT_READ  hipHostRegister(flags={name}): rc {rc} {lib.hipGetErrorString(rc).decode()}  {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.2f} GB/s")
                if rc == 0:
                    copy_rate(ptr, H2D, "H2D from the mapping")
                    got = dbuf[:1 << 20].cpu().numpy().tobytes() == blk
                    print(f"    bytes arrived intact: {got}")
                    t = time.perf_counter()
                    rc = lib.hipHostUnregister(ptr)
                    print(f"    hipHostUnregister: rc {rc}  {(time.perf_counter() - t) * 1e3:.1f} ms")
                    # per-batch registration
                    t = time.perf_counter()
                    k = 0
                    for off in range(0, nbytes - batch + 1, batch):
                        a0 = off // mmap.PAGESIZE * mmap.PAGESIZE
                        a1 = -(-(off + batch) // mmap.PAGESIZE) * mmap.PAGESIZE
                        if lib.hipHostRegister(ptr + a0, a1 - a0, flags):
                            print("    per-batch registration failed"); lib.hipGetLastError(); break
                        lib.hipMemcpyAsync(dbuf.data_ptr() + off, ptr + off, batch, H2D, stream)
                        torch.cuda.synchronize()
                        lib.hipHostUnregister(ptr + a0)
                        k += 1
                    dt = time.perf_counter() - t
                    if k:
                        print(f"    register + copy + unregister per 16-frame batch: {dt / k * 1e3:.1f} ms per batch = {k * batch / dt / 1e9:.1f} GB/s")
                    break
                lib.hipGetLastError()
            del arr
            m.close(); os.close(fd)
            # ---- output: ftruncate + writable shared mapping
            opath = path + ".out"
            fd = os.open(opath, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
            os.ftruncate(fd, nbytes)
            m = mmap.mmap(fd, nbytes, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ | mmap.PROT_WRITE)
            arr = np.frombuffer(m, dtype=np.uint8)
            ptr = arr.ctypes.data
            t = time.perf_counter()
            rc = lib.hipHostRegister(ptr, nbytes, 0)
            dt = time.perf_counter() - t
            print(f"  output PROT_WRITE hipHostRegister(default): rc {rc} {lib.hipGetErrorString(rc).decode()}  {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.2f} GB/s")
            if rc == 0:
                copy_rate(ptr, D2H, "D2H into the mapping")
                lib.hipHostUnregister(ptr)
                with open(opath, "rb") as f:
                    print(f"    file holds the bytes: {f.read(1 << 20) == blk}")
            else:
                lib.hipGetLastError()
            del arr
            m.close(); os.close(fd)
            # ---- today's staging on the same file: mmap -> pinned memcpy on 8 / 16 threads, and pwrite of a pinned batch
            from concurrent.futures import ThreadPoolExecutor
            fd = os.open(path, os.O_RDONLY)
            m = mmap.mmap(fd, nbytes, prot=mmap.PROT_READ)
            arr = np.frombuffer(m, dtype=np.uint8)
            pin = torch.empty(batch, dtype=torch.uint8).pin_memory().numpy()
            for nt in (8, 16):
                with ThreadPoolExecutor(nt) as ex:
                    sl = 8 << 20
                    t = time.perf_counter()
                    for off in range(0, nbytes - batch + 1, batch):
                        list(ex.map(lambda lo: np.copyto(pin[lo:lo + sl], arr[off + lo:off + lo + sl][:len(pin[lo:lo + sl])]), range(0, batch, sl)))
                    dt = time.perf_counter() - t
                    print(f"  memcpy mapping -> pinned, {nt} threads: {(nbytes // batch) * batch / dt / 1e9:.1f} GB/s")
            del arr
            m.close(); os.close(fd)
        finally:
            for p in (path, path + ".out"):
                try:
                    os.remove(p)
                except OSError:
                    pass


if __name__ == "__main__":
    main()
