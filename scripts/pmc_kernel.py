"""Average counter values per launch for kernels matching a substring, from a rocprofv3 counter_collection.csv.
usage: pmc_kernel.py <csv> <kernel-substring>"""
import collections
import csv
import sys

agg = collections.defaultdict(float)
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]].add(r["Dispatch_Id"])
for k, v in sorted(agg.items()):
    print(f"{k:32s} {v / max(len(disp[k]), 1):16.1f} per launch  ({len(disp[k])} launches)")
