#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_* (tools/collect_profiles.sh) into the tracked files under profiles/:
<tag>_kernel_stats.csv, <tag>_pmc.json, <tag>_bench.json and profiles/traffic.json (HBM bytes per frame
from the separate FETCH_SIZE / WRITE_SIZE passes; KiB -> bytes, FETCH_SIZE of the k_warp kernels doubled: see DESIGN.md 6)."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_z"
cfg = sys.argv[2] if len(sys.argv) > 2 else "config3"
os.makedirs("profiles", exist_ok=True)
newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]      # gpurun_out/ accumulates across calls
ks = newest(glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True))
if ks:
    rows = [r for r in csv.reader(open(ks[0]))]
    keep = [rows[0]] + [r for r in rows[1:] if "crtfx" in r[0]]
    csv.writer(open(f"profiles/{tag}_kernel_stats.csv", "w")).writerows(keep)
pmc = {}
for part in ("fetch", "write", "sq"):
    fs = newest(glob.glob(f"gpurun_out/{tag}_{part}/**/*counter_collection.csv", recursive=True))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"]
        if "crtfx" in k:
            agg[k.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        pmc.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
json.dump(pmc, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)
# frames per launch of each kernel class, from the bench line of the same collection (PMC values are per launch)
fpl = {}
if os.path.exists(f"gpurun_out/{tag}_bench.json"):
    rl = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1]).get("roofline", {})
    fpl = {k: v["frames_per_launch"] for k, v in rl.get("kernels", {}).items()}
cls = lambda name: "k_phosphor" if ("k_phosphor" in name or "k_point" in name or "k_half" in name) else "k_warp"
# gfx950: FETCH_SIZE tallies 128-B requests at 64 B.  Calibrated on this code's own patterns (profiles/r01_z_fetch_calibration.txt):
# the 12-B-per-lane reads of the k_warp kernels report 0.44 x a known byte count -> doubled; the byte loads of k_phosphor / k_point
# are uncalibrated and taken as reported; WRITE_SIZE is exact.
fetch_fix = lambda name: 2.0 if cls(name) == "k_warp" else 1.0
traffic = sum((d.get("FETCH_SIZE", 0) * fetch_fix(k) + d.get("WRITE_SIZE", 0)) * 1024 / fpl.get(cls(k), 1.0) for k, d in pmc.items())      # bytes per FRAME
tj = json.load(open("profiles/traffic.json")) if os.path.exists("profiles/traffic.json") else {}
tj[cfg] = int(traffic)
tj[cfg + "_detail_bytes_per_launch_as_reported"] = {k: dict({c: int(v * 1024) for c, v in d.items() if c in ("FETCH_SIZE", "WRITE_SIZE")},
                                             frames_per_launch=fpl.get(cls(k), 1.0)) for k, d in pmc.items()}
tj.pop(cfg + "_detail_bytes", None)
tj.pop(cfg + "_detail_bytes_per_launch", None)
json.dump(tj, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
if os.path.exists(f"gpurun_out/{tag}_bench.json"):
    shutil.copy(f"gpurun_out/{tag}_bench.json", f"profiles/{tag}_bench.json")
print(json.dumps(tj, indent=1))
print(open(f"profiles/{tag}_kernel_stats.csv").read() if ks else "no kernel stats")
