"""per-kernel sums of the counters of rocprofv3 --pmc passes: python pmc_kernels.py <dir> [<dir> ...]"""
import csv, glob, sys, re, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void |\(anonymous namespace\)::|\(.*", "", r["Kernel_Name"])[:30]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in tot.items():
    print(k, {a: "%.3g" % b for a, b in sorted(c.items())})
