#!/usr/bin/env python3
"""Aggregator of tools/pmc_vmem.sh:  pmc_vmem_aggregate.py <out dir> <tag> "<pass>|<counters>" ...

Reads <out>/<tag>_<pass>/**/*counter_collection.csv of every named pass, averages each counter per crtfx kernel over its dispatches and writes
<out>/<tag>_vmem.json — ONLY when every counter of every pass is present for every crtfx kernel seen in any pass.  A missing pass or counter is
an error (exit 2, the missing names printed): a thinner JSON is never written (round 5 lost its `tc` counters that way, unnoticed)."""
import collections
import csv
import glob
import json
import os
import sys


def aggregate(out_dir, tag, passes):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    want = []
    problems = []
    for spec in passes:
        name, counters = spec.split("|", 1)
        want += counters.split()
        files = glob.glob(os.path.join(out_dir, f"{tag}_{name}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            problems.append(f"pass '{name}': no counter_collection.csv under {tag}_{name}/")
        for f in files:
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    if "crtfx" in k:
                        agg[k.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if not agg and not problems:
        problems.append("no crtfx kernel in any pass")
    for k, d in agg.items():
        missing = [c for c in want if c not in d]
        if missing:
            problems.append(f"{k}: missing {' '.join(missing)}")
    out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
    return out, problems


def main(argv):
    out_dir, tag, passes = argv[1], argv[2], argv[3:]
    out, problems = aggregate(out_dir, tag, passes)
    if problems:
        for p in problems:
            print("pmc_vmem_aggregate:", p, file=sys.stderr)
        print(f"pmc_vmem_aggregate: {tag}_vmem.json NOT written", file=sys.stderr)
        return 2
    with open(os.path.join(out_dir, f"{tag}_vmem.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, d in out.items():
        print(k)
        for c in sorted(d):
            print(f"   {c:36s} {d[c]:16.0f}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
