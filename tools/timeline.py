#!/usr/bin/env python3
"""Text timeline of a rocprofv3 --kernel-trace --output-format csv run: start, end, duration (ms), hardware queue,
kernel, grid — the last `--last` ms of the run, kernels shorter than `--min` ms left out.
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [--last 900] [--min 0.3]"""
import csv, glob, os, re, sys


def main():
    args = sys.argv[1:]
    last, tmin = 900.0, 0.3
    if "--last" in args:
        last = float(args[args.index("--last") + 1])
    if "--min" in args:
        tmin = float(args[args.index("--min") + 1])
    files = glob.glob(os.path.join(args[0], "**", "*_kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(files[0])))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    tend = (max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e6
    print("# start_ms end_ms duration_ms queue kernel grid (threads x rows) workgroup vgprs")
    for r in rows:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        if e < tend - last or e - s < tmin:
            continue
        name = re.sub(r"^void |\(anonymous namespace\)::|\(.*", "", r["Kernel_Name"])[:28]
        print("%9.2f %9.2f %7.2f q%s %-28s %sx%s wg %s vgpr %s" % (s, e, e - s, r["Queue_Id"], name, r["Grid_Size_X"], r["Grid_Size_Y"],
                                                                 r["Workgroup_Size_X"], r["VGPR_Count"]))


if __name__ == "__main__":
    main()
