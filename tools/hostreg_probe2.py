"""Dev tool (GPU box), second pass of hostreg_probe.py: per-BATCH cost of the zero-copy path on FRESH mappings, as the CLI would run it —
input: parallel MADV_POPULATE_READ of a window, hipHostRegister, H2D, unregister; output: parallel MADV_POPULATE_WRITE of a window of a
just-ftruncated file, register, D2H, unregister.     python tools/hostreg_probe2.py  ->  stdout"""
import ctypes
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

POP_READ, POP_WRITE = 22, 23


def hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    return ctypes.CDLL("libamdhip64.so")


def main():
    frame = 3840 * 2160 * 3
    batch = frame * 16
    nb = 6
    nbytes = batch * nb
    dev = torch.device("cuda", 0)
    dbuf = torch.empty(batch, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    lib = hip()
    lib.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    lib.hipHostUnregister.argtypes = [ctypes.c_void_p]
    lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    stream = torch.cuda.current_stream(dev).cuda_stream
    blk = np.random.default_rng(0).integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    for d in ("/dev/shm", "/tmp"):
        path = os.path.join(d, "crtfx_probe2.bin")
        with open(path, "wb") as f:
            for _ in range(nbytes // len(blk) + 1):
                f.write(blk)
        print(f"{d}: {nb} batches of 16 4K frames ({batch >> 20} MiB)")
        for nt in (4, 8, 16):
            for populate in (True, False):
                fd = os.open(path, os.O_RDONLY)
                m = mmap.mmap(fd, nbytes, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
                ptr = np.frombuffer(m, dtype=np.uint8).ctypes.data
                tp = tr = tc = 0.0
                with ThreadPoolExecutor(nt) as ex:
                    for b in range(nb):
                        off = b * batch
                        t = time.perf_counter()
                        if populate:
                            sl = -(-batch // nt // 4096) * 4096
                            list(ex.map(lambda lo: m.madvise(POP_READ, off + lo, min(sl, batch - lo)), range(0, batch, sl)))
                        tp += time.perf_counter() - t
                        t = time.perf_counter()
                        rc = lib.hipHostRegister(ptr + off, batch, 0)
                        tr += time.perf_counter() - t
                        if rc:
                            print("   register failed", rc); break
                        t = time.perf_counter()
                        lib.hipMemcpyAsync(dbuf.data_ptr(), ptr + off, batch, 1, stream)
                        torch.cuda.synchronize()
                        tc += time.perf_counter() - t
                        lib.hipHostUnregister(ptr + off)
                print(f"  input  {nt:2d} threads populate={populate!s:5}: populate {tp / nb * 1e3:6.2f} ms  register {tr / nb * 1e3:6.2f} ms  H2D {tc / nb * 1e3:6.2f} ms per batch"
                      f"  -> serial {batch / ((tp + tr + tc) / nb) / 1e9:5.1f} GB/s, host part alone {batch / max(1e-9, (tp + tr) / nb) / 1e9:6.1f} GB/s")
                m.close(); os.close(fd)
            for mode in ("populate", "fallocate", "none"):
                opath = path + ".out"
                fd = os.open(opath, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
                os.ftruncate(fd, nbytes)
                m = mmap.mmap(fd, nbytes, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ | mmap.PROT_WRITE)
                ptr = np.frombuffer(m, dtype=np.uint8).ctypes.data
                tp = tr = tc = 0.0
                with ThreadPoolExecutor(nt) as ex:
                    for b in range(nb):
                        off = b * batch
                        sl = -(-batch // nt // 4096) * 4096
                        t = time.perf_counter()
                        if mode == "populate":
                            list(ex.map(lambda lo: m.madvise(POP_WRITE, off + lo, min(sl, batch - lo)), range(0, batch, sl)))
                        elif mode == "fallocate":
                            list(ex.map(lambda lo: os.posix_fallocate(fd, off + lo, min(sl, batch - lo)), range(0, batch, sl)))
                        tp += time.perf_counter() - t
                        t = time.perf_counter()
                        rc = lib.hipHostRegister(ptr + off, batch, 0)
                        tr += time.perf_counter() - t
                        if rc:
                            print("   register failed", rc); break
                        t = time.perf_counter()
                        lib.hipMemcpyAsync(ptr + off, dbuf.data_ptr(), batch, 2, stream)
                        torch.cuda.synchronize()
                        tc += time.perf_counter() - t
                        lib.hipHostUnregister(ptr + off)
                print(f"  output {nt:2d} threads {mode:9}: prepare {tp / nb * 1e3:6.2f} ms  register {tr / nb * 1e3:6.2f} ms  D2H {tc / nb * 1e3:6.2f} ms per batch"
                      f"  -> serial {batch / ((tp + tr + tc) / nb) / 1e9:5.1f} GB/s, host part alone {batch / max(1e-9, (tp + tr) / nb) / 1e9:6.1f} GB/s")
                m.close(); os.close(fd); os.remove(opath)
            # today's output path for comparison: pwrite of a pinned batch in parallel slices
            pin = torch.empty(batch, dtype=torch.uint8).pin_memory()
            view = memoryview(pin.numpy())
            fd = os.open(path + ".out", os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
            with ThreadPoolExecutor(nt) as ex:
                t = time.perf_counter()
                for b in range(nb):
                    sl = 8 << 20
                    list(ex.map(lambda lo: os.pwrite(fd, view[lo:lo + sl], b * batch + lo), range(0, batch, sl)))
                dt = time.perf_counter() - t
            print(f"  output {nt:2d} threads pwrite of a pinned batch: {dt / nb * 1e3:6.2f} ms per batch = {batch * nb / dt / 1e9:.1f} GB/s")
            os.close(fd); os.remove(path + ".out")
        os.remove(path)


if __name__ == "__main__":
    main()
