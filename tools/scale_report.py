#!/usr/bin/env python3
"""Read bench.py lines taken at N = 1, 2, 4, 8 GPUs and say, from the lines alone, how the frame-sharded path scaled and — when it is under
the target — where the loss went.  The run sheet of DESIGN.md section 7 as a program: for whoever takes the 8-GPU measurement, once.

    python tools/scale_report.py line_n1.json line_n2.json ... [--target 7.0]
    python bench.py --gpus 8 --config 4 | python tools/scale_report.py line_n1.json -        ('-' = a line on stdin)

Every file holds one bench.py JSON line (other lines are ignored), or a driver record with the line under "parsed".  No GPU, no torch.
SURVEY 8e / north_star: ">= 7x frames/sec at 8 GPUs vs 1 GPU on the frame-sharded path" (crt_filter.py ref:1015-1017, :1081-1105 is the
strategy being scaled: frames in parallel, in-order persistence commit)."""
import json
import sys


def load_line(path):
    text = sys.stdin.read() if path == "-" else open(path).read()
    try:
        d = json.loads(text)
        if isinstance(d, dict) and "parsed" in d and isinstance(d["parsed"], dict):
            d = d["parsed"]
        if isinstance(d, dict) and "value" in d:
            return d
    except ValueError:
        pass
    for ln in text.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    raise SystemExit(f"{path}: no bench.py line found")


def diagnose(base, line, target_eff=0.875):
    """(verdict, [findings]) for one N > 1 line against the N = 1 line of the same config."""
    n = int(line["n_gpus"])
    speedup = line["value"] / base["value"]
    eff = speedup / n
    out = [f"N = {n}: {line['value']:.0f} frames/s = {speedup:.2f}x of N = 1 ({base['value']:.0f}), efficiency {eff:.3f}"]
    d = line.get("dist") or {}
    if d.get("world_size_seen") != n:
        out.append(f"  !! dist.world_size_seen = {d.get('world_size_seen')}, expected {n}: not the run it claims to be")
    pr = [r for r in (d.get("per_rank") or []) if isinstance(r, dict)]
    if len(pr) != n:
        out.append(f"  !! {len(pr)} per-rank records for {n} ranks")
    base_gpu = ((base.get("dist") or {}).get("per_rank") or [{}])[0].get("gpu") or {}
    b_clk, b_pw = base_gpu.get("sclk_mhz_mean"), base_gpu.get("power_w_mean")
    clks = [r["gpu"].get("sclk_mhz_mean") for r in pr if r.get("gpu") and r["gpu"].get("sclk_mhz_mean")]
    pws = [r["gpu"].get("power_w_mean") for r in pr if r.get("gpu") and r["gpu"].get("power_w_mean")]
    own = [r.get("frames_per_s_own_clock") for r in pr if r.get("frames_per_s_own_clock")]
    chain = [r.get("chain_ms_per_frame") for r in pr if r.get("chain_ms_per_frame")]
    causes = []
    if clks and b_clk:
        lo, hi = min(clks), max(clks)
        out.append(f"  shader clock per rank {lo:.0f} .. {hi:.0f} MHz (N = 1: {b_clk:.0f}); package power {min(pws):.0f} .. {max(pws):.0f} W (N = 1: {b_pw:.0f})" if pws and b_pw
                   else f"  shader clock per rank {lo:.0f} .. {hi:.0f} MHz (N = 1: {b_clk:.0f})")
        if lo < 0.97 * b_clk:
            causes.append(f"clock: the slowest rank runs {lo / b_clk:.3f} of the 1-GPU clock -> a node power / thermal cap (the chain's frame rate follows the shader clock)")
        elif hi - lo > 0.03 * b_clk:
            causes.append("clock: ranks differ by more than 3 % -> uneven cooling / power sharing")
    sched = line.get("shard_schedule")
    if sched:
        tot = sched.get("scan_us", 0) + sched.get("hop_stall_us", 0) + sched.get("fixup_us", 0)
        out.append(f"  persistence hop ({d.get('hop_schedule')}): scan {sched.get('scan_us', 0) / 1e3:.1f} ms, hop stall {sched.get('hop_stall_us', 0):.0f} us, "
                   f"fix-up {sched.get('fixup_us', 0):.0f} us per round = {sched.get('hop_plus_fixup_share', 0):.4f} of the round")
        if tot and sched.get("hop_plus_fixup_share", 0) > 0.02:
            causes.append(f"hop: hop + fix-up take {sched['hop_plus_fixup_share']:.3f} of a round (healthy: < 0.02) -> the xGMI link or RCCL's stream; try CRTFX_SHARD_OVERLAP=1")
    if own:
        lo, hi = min(own), max(own)
        out.append(f"  frames/s on each rank's own clock {lo:.0f} .. {hi:.0f}")
        if chain and max(chain) - min(chain) < 0.03 * max(chain) and lo < 0.93 * hi:
            causes.append("straggler: one rank is slower on its own clock with the same kernel time per frame -> its host side (CPU affinity / a busy core)")
        if lo * n > 1.05 * line["value"]:
            causes.append("start skew: every rank is fast on its own clock but the max-over-ranks region is longer -> ranks entered the timed region late (barrier / launch skew)")
    if chain:
        b_chain = ((base.get("dist") or {}).get("per_rank") or [{}])[0].get("chain_ms_per_frame")
        if b_chain and max(chain) > 1.05 * b_chain and not any(c.startswith("clock") for c in causes):
            causes.append(f"kernels: {max(chain) * 1e3:.1f} us per frame against {b_chain * 1e3:.1f} at N = 1 with equal clocks -> shared HBM / fabric contention is not expected "
                          "on this path (every rank has its own stack): look at the box")
    verdict = "ok" if eff >= target_eff else "UNDER TARGET"
    if eff < target_eff and not causes:
        causes.append("no single cause visible in the line: compare `ms_per_step` x steps with `timed_region_s`, and the host_tables share")
    for c in causes:
        out.append("  -> " + c)
    return verdict, out


def main(argv):
    target = 7.0
    paths = []
    it = iter(argv[1:])
    for a in it:
        if a == "--target":
            target = float(next(it))
        else:
            paths.append(a)
    if not paths:
        print(__doc__)
        return 2
    lines = sorted((load_line(p) for p in paths), key=lambda d: int(d["n_gpus"]))
    base = next((d for d in lines if int(d["n_gpus"]) == 1), None)
    if base is None:
        raise SystemExit("an N = 1 line of the same config is needed as the base")
    wl = base["config"]["workload"]
    print(f"workload: {wl}")
    bad = 0
    for d in lines:
        if d is base:
            g = ((d.get("dist") or {}).get("per_rank") or [{}])[0].get("gpu") or {}
            print(f"N = 1: {d['value']:.0f} frames/s, shader clock {g.get('sclk_mhz_mean')} MHz, package {g.get('power_w_mean')} W")
            continue
        if d["config"]["workload"] != wl:
            raise SystemExit(f"N = {d['n_gpus']}: another workload ({d['config']['workload']})")
        n = int(d["n_gpus"])
        verdict, out = diagnose(base, d, target_eff=(target / 8.0))
        print("\n".join(out))
        if n == 8:
            ok = d["value"] / base["value"] >= target
            print(f"  8-GPU target {target:.1f}x: {'MET' if ok else 'NOT MET'} ({d['value'] / base['value']:.2f}x)")
            bad += 0 if ok else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
