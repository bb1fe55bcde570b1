#!/usr/bin/env python
"""Per-kernel averages of the rocprofv3 --pmc passes tools/prof_kernel.sh wrote under gpurun_out/<dir> (attention kernels only)."""
import collections, csv, glob, sys
d = sys.argv[1]
for p in ("pmc1", "pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, p), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("void latte::(anonymous namespace)::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kname, c in acc.items():
        if "attn" not in kname:
            continue
        print(kname)
        for n, v in sorted(c.items()):
            print("   %-28s %14.0f (n=%d)" % (n, sum(v) / len(v), len(v)))
