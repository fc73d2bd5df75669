"""Host <-> device copy rates of this box through pinned memory (what the CLI's staging can reach): each direction alone and both at
once on two streams.    python tools/pcie_rates.py [MB]"""
import sys
import time

import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 398        # 16 frames of 4K rgb24
n = mb << 20
dev = torch.device("cuda", 0)
h_in, h_out = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
d_in, d_out = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def up():
    with torch.cuda.stream(s1):
        d_in.copy_(h_in, non_blocking=True)


def down():
    with torch.cuda.stream(s2):
        h_out.copy_(d_out, non_blocking=True)


tu, td = timed(up), timed(down)
tb = timed(lambda: (up(), down()))
print(f"{mb} MiB pinned: H2D {n / tu / 1e9:.1f} GB/s, D2H {n / td / 1e9:.1f} GB/s, both at once {n / tb / 1e9:.1f} GB/s each way "
      f"({2 * n / tb / 1e9:.1f} GB/s total; serialised they would take {tu + td:.4f} s, together {tb:.4f} s)")
print(f"4K rgb24 frames/s this allows: one direction at a time {n / (tu + td) / 24883200:.0f}, both directions concurrent {n / tb / 24883200:.0f}")
