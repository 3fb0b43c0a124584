#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch.
usage: python tools/pmc_summary.py gpurun_out/pmc_dir [more dirs...]"""
import csv, glob, os, sys, collections

def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows

def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0]

def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        per_dispatch = collections.defaultdict(dict)
        for r in load(d):
            key = (r["Dispatch_Id"], short(r["Kernel_Name"]), r.get("Grid_Size", ""))
            per_dispatch[key][r["Counter_Name"]] = per_dispatch[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (did, k, g), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[(k, g)][c].append(v)
    for (k, g) in sorted(acc):
        if not any(x in k for x in ("k_rollout", "k_backward", "k_derivs")):
            continue
        print("%s grid=%s" % (k, g))
        for c in sorted(acc[(k, g)]):
            v = acc[(k, g)][c]
            print("    %-32s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))

if __name__ == "__main__":
    main()
