#!/usr/bin/env python3
"""Weight the VALU instructions of a kernel's main loop by their measured gfx950 issue cost (tools/ubench/valu_cost.hip,
profiles/r02_valu_cost.txt: shader cycles per wave-instruction at a saturated SIMD, 4 waves/SIMD).  The sum is the
SIMD time one wave's trip through the loop needs; x waves per SIMD / trips gives the kernel's VALU floor.

    hipcc ... -S --cuda-device-only -o rr9.s crtfx_rr.hip
    python tools/isa_cost.py rr9.s '<mangled kernel name prefix>' [first_line last_line]
"""
import re, sys, collections
COST = {  # cycles per wave-instruction, 4 waves/SIMD (measured), by mnemonic prefix; first match wins
    "v_pk_fma_f32": 3.4, "v_pk_mul_f32": 3.35, "v_pk_add_f32": 3.35, "v_pk_mov_b32": 3.3,
    "v_fmac_f32": 2.4, "v_fma_f32": 2.0, "v_mul_f32": 1.95, "v_add_f32": 1.95, "v_sub_f32": 1.95, "v_subrev_f32": 1.95,
    "v_max_f32": 2.2, "v_min_f32": 2.2, "v_med3_f32": 2.9, "v_rndne_f32": 2.2,
    "v_mul_f64": 2.9, "v_add_f64": 3.3, "v_fma_f64": 3.3, "v_max_f64": 3.3, "v_min_f64": 3.3,
    "v_cvt_": 3.7, "v_log_f32": 5.8, "v_sqrt_f32": 5.8, "v_cos_f32": 5.8, "v_sin_f32": 5.8, "v_rcp_f32": 5.8, "v_exp_f32": 5.8, "v_rsq_f32": 5.8,
    "v_lshl_add_u32": 3.4, "v_mad_u32_u24": 3.4, "v_mad_u64_u32": 6.8, "v_bfe_u32": 3.4, "v_mul_lo_u32": 3.4, "v_mul_hi_u32": 3.4, "v_add3_u32": 3.4,
    "v_lshl_or_b32": 3.4, "v_and_or_b32": 3.4, "v_or3_b32": 3.4, "v_xad_u32": 3.4, "v_med3_i32": 3.4, "v_lshl_add_u64": 3.4, "v_alignbit": 3.4, "v_perm_b32": 3.4,
    "v_cndmask_b32": 2.0, "v_mov_b32": 1.95, "v_readlane": 4.0, "v_writelane": 4.0, "v_readfirstlane": 4.0,
    "v_cmp": 2.0, "v_": 1.95,
}
def cost(m):
    for k, v in COST.items():
        if m.startswith(k):
            return v
    return 0.0
def static_mix(path):
    """{mangled kernel name: (VALU instructions, weighted VALU cycles)} over the whole body of every function of an ISA listing
    (hipcc -S --cuda-device-only): the static instruction mix, used by tools/summarise_profiles.py to turn SQ_INSTS_VALU into
    cost-weighted SIMD cycles (average cycles per VALU wave-instruction of that kernel)."""
    out, name, n, cyc = {}, None, 0, 0.0
    for l in open(path).read().splitlines():
        lab = re.match(r"^([A-Za-z_$][\w$.]*):", l)       # "name:   ; @name" — local labels start with '.'
        if lab:
            name, n, cyc = lab.group(1), 0, 0.0
            continue
        if l.startswith(".Lfunc_end") and name:
            out[name] = (n, cyc)
            name = None
            continue
        if name is None:
            continue
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        if t[0].startswith("v_"):
            n += 1
            cyc += cost(t[0])
    return out
def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().endswith(":") or (l.startswith(name) and ":" in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
    hist, cyc = collections.Counter(), collections.Counter()
    n = 0
    for l in body[lo:hi]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        m = t[0]
        n += 1
        hist[m] += 1
        cyc[m] += cost(m)
    tot = sum(cyc.values())
    print(f"{name[:60]} lines {lo}..{hi}: {n} instructions, VALU {sum(v for k, v in hist.items() if k.startswith('v_'))}, weighted VALU cycles {tot:.0f}")
    cls = collections.Counter()
    for m, c in cyc.items():
        key = ("fma/pk" if "fma" in m else "f64" if "f64" in m and "cvt" not in m else "cvt" if "cvt" in m else "trans" if m in ("v_log_f32", "v_sqrt_f32", "v_cos_f32", "v_sin_f32", "v_rcp_f32") else
               "int/addr" if re.search(r"_(u32|i32|b32|u64|u24|b64|u16)", m) else "f32 other")
        cls[key] += c
    for k, v in cls.most_common():
        print(f"   {k:10s} {v:7.0f} cyc  {100 * v / tot:5.1f} %")
    for m, c in cyc.most_common(22):
        print(f"      {m:22s} x{hist[m]:4d}  {c:6.0f}")
    other = {m: h for m, h in hist.items() if not m.startswith("v_")}
    print("   non-VALU:", ", ".join(f"{m} x{h}" for m, h in sorted(other.items(), key=lambda kv: -kv[1])[:14]))
if __name__ == "__main__":
    main()
