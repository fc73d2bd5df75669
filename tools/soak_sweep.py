"""More of tests/test_sweep_gpu.py than the test suite runs: the same seeded random render-loop cases against the oracle for any range of
case numbers, optionally at larger frame sizes (wide enough for interior strips, widths that are / are not a multiple of 4 — the branch-free
warp builds need whole dwords per row).    python tools/soak_sweep.py [first last] [--big | --half]      (GPU box; a one-off check, not part of pytest)
--half: the float16 sweep of the column-owner kernel (test_random_full_chain_on_the_half_kernel) instead."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_sweep_gpu as sw      # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
first, last = (int(args[0]), int(args[1])) if len(args) >= 2 else (150, 900)
if "--big" in sys.argv:
    sw.SIZES = [(270, 480), (203, 264), (96, 512), (120, 324), (64, 702), (151, 330)]
bad, t0 = 0, time.time()
for case in range(first, last):
    try:
        (sw.test_random_full_chain_on_the_half_kernel if "--half" in sys.argv else sw.test_random_render_matches_oracle)(case)
    except AssertionError as e:
        bad += 1
        print("FAIL case", case, str(e)[:300], flush=True)
    if (case - first) % 50 == 49:
        print(f"case {case}: {time.time() - t0:.1f} s, {bad} failures so far", flush=True)
print(f"done: cases {first}..{last - 1} at sizes {sw.CT_SIZES if '--half' in sys.argv else sw.SIZES}{' (float16 frames)' if '--half' in sys.argv else ''}: {bad} failures")
sys.exit(1 if bad else 0)
