#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel of a gfx950 ISA listing (hipcc -S --cuda-device-only): instructions, VALU
(and their cost-weighted cycles, tools/isa_cost.py), LDS, vector / scalar memory, scratch (spill) traffic, barriers, waits.
The loop bodies of k_phosphor_* are the blocks holding an s_barrier.

    python tools/isa_blocks.py rr9.s k_phosphor_ct [min_instructions]
"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_cost import cost


def blocks(path, needle):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if needle in l.split(":")[0] and l and not l[0].isspace() and ":" in l and not l.startswith((".", ";")))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur, out = "entry", {}
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = m.group(1)
            continue
        if not t or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        s = out.setdefault(cur, dict(n=0, valu=0, cyc=0.0, ds=0, vmem=0, smem=0, scratch=0, barrier=0, waitcnt=0, pk=0))
        s["n"] += 1
        if op.startswith("scratch_"): s["scratch"] += 1
        elif op.startswith("ds_"): s["ds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_")): s["vmem"] += 1
        elif op.startswith(("s_load", "s_buffer_load")): s["smem"] += 1
        elif op == "s_barrier": s["barrier"] += 1
        elif op == "s_waitcnt": s["waitcnt"] += 1
        if op.startswith("v_"):
            s["valu"] += 1
            s["cyc"] += cost(op)
            if op.startswith("v_pk_fma"): s["pk"] += 1
    return out


if __name__ == "__main__":
    path, needle = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    tot = dict()
    for b, s in blocks(path, needle).items():
        if s["n"] >= lo:
            print(f"{b:12s} n {s['n']:5d}  valu {s['valu']:5d} ({s['cyc']:7.0f} cyc, pk_fma {s['pk']:3d})  ds {s['ds']:3d}  vmem {s['vmem']:3d}  smem {s['smem']:3d}  "
                  f"scratch {s['scratch']:3d}  barrier {s['barrier']}  waitcnt {s['waitcnt']:3d}")
