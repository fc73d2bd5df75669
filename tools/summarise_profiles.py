#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_* (tools/collect_profiles.sh) into the tracked files under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats rows of the crtfx kernels
  <tag>_pmc.json           separate --pmc passes, mean per launch
  <tag>_bench.json         the bench line of the same box
  <tag>_fetch_calibration.txt   FETCH_SIZE on a known cold byte count in the chain's load shapes
  traffic.json / valu.json what bench.py's roofline.traffic / roofline.valu read — each entry carries the sha1 of the kernel
                           sources it was measured on (bench.py ignores it when that differs from the build under test), the
                           tag, the batch, and BOTH the as-reported and the corrected byte counts.
Usage: python tools/summarise_profiles.py <tag> [configN]"""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_z"
cfg = sys.argv[2] if len(sys.argv) > 2 else "config3"
os.makedirs("profiles", exist_ok=True)
newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]      # gpurun_out/ accumulates across calls


def counters(part):
    agg_file = f"gpurun_out/{tag}_{part}_agg.json"          # per-kernel means made on the GPU box (tools/collect_profiles.sh); else the raw rows
    if os.path.exists(agg_file):
        return json.load(open(agg_file))
    fs = newest(glob.glob(f"gpurun_out/{tag}_{part}/**/*counter_collection.csv", recursive=True))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in (csv.DictReader(open(fs[0])) if fs else []):
        agg[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


try:
    pmc_batch = int(open(f"gpurun_out/{tag}_pmc_batch.txt").read().strip())      # frames per step of the PMC passes (tools/collect_profiles.sh)
except (OSError, ValueError):
    pmc_batch = 8
# (bench.py runs tools/ubench/mall_copy as a child process: under rocprofv3 that process leaves a kernel_stats.csv of its own — take the one with crtfx kernels)
ks = newest([f for f in glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True) if "crtfx" in open(f).read()])
if ks:
    rows = [r for r in csv.reader(open(ks[0]))]
    csv.writer(open(f"profiles/{tag}_kernel_stats.csv", "w")).writerows([rows[0]] + [r for r in rows[1:] if "crtfx" in r[0]])
pmc = {}
for part in ("fetch", "write", "sq", "sq2"):
    for k, d in counters(part).items():
        if "crtfx" in k:
            pmc.setdefault(k, {}).update(d)
json.dump(pmc, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)

# ---- FETCH_SIZE calibration: reported KiB vs the bytes each shape really read (2 GiB, cold) --------------------------------
calib = {k: d.get("FETCH_SIZE", 0.0) * 1024 for k, d in counters("calib").items()}
known = {}
log = f"gpurun_out/{tag}_calib.log"
if os.path.exists(log):
    m = re.search(r"k_px12 (\d+)\s+k_byte3 (\d+)\s+k_dword (\d+)", open(log).read())
    if m:
        known = {"k_px12": int(m.group(1)), "k_byte3": int(m.group(2)), "k_dword": int(m.group(3))}
ratio = {k: (calib[n] / known[k]) for k in known for n in calib if n.startswith(k) and known[k]}
with open(f"profiles/{tag}_fetch_calibration.txt", "w") as f:
    f.write("FETCH_SIZE (rocprofv3 --pmc, KiB -> bytes) against a known byte count read ONCE from a cold 2 GiB buffer (8x the Infinity Cache)\n"
            "tools/ubench/fetch_calib.hip; shapes: k_px12 = 12 B/lane float3 pixels (k_warp taps), k_byte3 = 3 single-byte loads per lane at a\n"
            "3-byte lane stride (k_phosphor frame bytes), k_dword = 4 B/lane.\n")
    for k in known:
        n = next((x for x in calib if x.startswith(k)), None)
        if n:
            f.write(f"  {k:8s} read {known[k] / 1e6:9.1f} MB   FETCH_SIZE reports {calib[n] / 1e6:9.1f} MB   ratio {ratio[k]:.3f}\n")
    f.write("bench.py's roofline.traffic divides a kernel's reported FETCH_SIZE by the ratio of its load shape (a ratio within 5 % of 1 is taken as 1).\n")
fix = lambda r: 1.0 if (r is None or abs(r - 1.0) < 0.05 or r <= 0) else 1.0 / r
fetch_fix = {"k_warp": fix(ratio.get("k_px12")), "k_phosphor": fix(ratio.get("k_byte3")), "k_bloom_pass": fix(ratio.get("k_byte3"))}

# ---- frames per launch of each kernel class, from the bench line of the same collection (PMC values are per launch) ------
fpl, bench = {}, {}
if os.path.exists(f"gpurun_out/{tag}_bench.json"):
    bench = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
    fpl = {k: v["frames_per_launch"] for k, v in bench.get("roofline", {}).get("kernels", {}).items()}
    shutil.copy(f"gpurun_out/{tag}_bench.json", f"profiles/{tag}_bench.json")
cls = lambda name: "k_bloom_pass" if ("k_half" in name or "k_sb_" in name) else "k_phosphor" if ("k_phosphor" in name or "k_point" in name) else "k_warp"
import bench as bench_mod      # source_hash(): sha1 of the kernel sources
src = bench_mod.source_hash()
raw = sum((d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 / fpl.get(cls(k), 1.0) for k, d in pmc.items())
cor = sum((d.get("FETCH_SIZE", 0) * fetch_fix[cls(k)] + d.get("WRITE_SIZE", 0)) * 1024 / fpl.get(cls(k), 1.0) for k, d in pmc.items())
tj = json.load(open("profiles/traffic.json")) if os.path.exists("profiles/traffic.json") else {}
tj = {k: v for k, v in tj.items() if isinstance(v, dict) and "source_hash" in v}      # drop round-1 style entries
tj[cfg] = {"bytes_per_frame_as_reported": int(raw), "bytes_per_frame_corrected": int(cor), "source_hash": src, "tag": tag, "batch": pmc_batch,
           "correction": {"FETCH_SIZE multiplier by kernel class (1 / calibration ratio)": fetch_fix, "WRITE_SIZE": "exact",
                          "calibration": f"profiles/{tag}_fetch_calibration.txt"},
           "per_launch_as_reported": {k: dict({c: int(v * 1024) for c, v in d.items() if c in ("FETCH_SIZE", "WRITE_SIZE")},
                                              frames_per_launch=fpl.get(cls(k), 1.0)) for k, d in pmc.items()}}
json.dump(tj, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
# VALU wave-instructions per frame, cost-weighted by the static instruction mix of each kernel's ISA, the LDS pipe's share, and
# how the dominant kernel's waves spend their time
import subprocess, tempfile
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_cost
from pythoncrt_amd import _lib as _build


def kernel_avg_cost():
    """{demangled kernel name: average measured issue cycles per VALU wave-instruction} for the kernels the PMC passes saw."""
    R = next((int(m.group(1)) for k in pmc for m in [re.search(r"k_phosphor_(?:cc|ct|rr)<(\d+)", k)] if m), 9)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for src, extra in (("crtfx_rr.hip", [f"-DRR_R={R}"]), ("crtfx.hip", [])):
            asm = os.path.join(td, src + ".s")
            subprocess.run(["hipcc", *_build.HIPCC_FLAGS, "-I", os.path.join(ROOT, "include"), "-I", _build.CSRC, *extra, "-S", "--cuda-device-only",
                            os.path.join(_build.CSRC, src), "-o", asm], check=True)
            mix = isa_cost.static_mix(asm)
            names = subprocess.run(["c++filt"], input="\n".join(mix), capture_output=True, text=True).stdout.splitlines()
            for mangled, dem in zip(mix, names):
                n, cyc = mix[mangled]
                if n:
                    out[dem.split("(")[0].replace("void ", "")] = cyc / n
    return out


avg = kernel_avg_cost()
cost_of = lambda k: avg.get(k) or next((v for n, v in avg.items() if n.startswith(k.split("<")[0])), 2.9)
valu = sum(d.get("SQ_INSTS_VALU", 0) / fpl.get(cls(k), 1.0) for k, d in pmc.items())
valu_w = sum(d.get("SQ_INSTS_VALU", 0) * cost_of(k) / fpl.get(cls(k), 1.0) for k, d in pmc.items())
lds_act = sum(d.get("SQ_LDS_IDX_ACTIVE", 0) / fpl.get(cls(k), 1.0) for k, d in pmc.items())
lds_bc = sum(d.get("SQ_LDS_BANK_CONFLICT", 0) / fpl.get(cls(k), 1.0) for k, d in pmc.items())
domk = max(pmc, key=lambda k: (pmc[k].get("GRBM_GUI_ACTIVE", 0), pmc[k].get("SQ_WAVE_CYCLES", 0))) if pmc else None      # the kernel that runs longest per launch
dom_wave = dom_lds = None
if domk:
    d = pmc[domk]
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1.0
    dom_wave = {"kernel": domk, "parked_SQ_WAIT_ANY": round(d.get("SQ_WAIT_ANY", 0) / wc, 3), "issue_stalled_SQ_WAIT_INST_ANY": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                "issuing_SQ_ACTIVE_INST_ANY": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}
    if d.get("GRBM_GUI_ACTIVE"):
        dom_lds = round(d.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * d["GRBM_GUI_ACTIVE"] / 8), 4)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
vj = json.load(open("profiles/valu.json")) if os.path.exists("profiles/valu.json") else {}
vj[cfg] = {"valu_wave_insts_per_frame": int(valu), "valu_cost_weighted_cycles_per_frame": int(valu_w),
           "valu_avg_cycles_per_inst": {k: round(cost_of(k), 3) for k in pmc},
           "lds_idx_active_cycles_per_frame": int(lds_act), "lds_bank_conflict_cycles_per_frame": int(lds_bc),
           "dominant_lds_pipe_frac": dom_lds, "dominant_wave_time": dom_wave,
           "clock_ghz": 2.4, "source_hash": src, "tag": tag,
           "cost_model": "SQ_INSTS_VALU x the kernel's average issue cycles per VALU wave-instruction: static mix of its ISA (tools/isa_cost.py) priced with profiles/r02_valu_cost.txt",
           "per_launch": {k: {c: d[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                                "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if c in d} for k, d in pmc.items()}}
json.dump(vj, open("profiles/valu.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_launch_as_reported"} for k, v in tj.items()}, indent=1))
print(open(f"profiles/{tag}_kernel_stats.csv").read() if ks else "no kernel stats")
print(open(f"profiles/{tag}_fetch_calibration.txt").read())
