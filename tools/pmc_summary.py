#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv files under a directory: mean per dispatch per kernel."""
import collections, csv, glob, os, sys
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    rows = list(csv.DictReader(open(f)))
    big = collections.defaultdict(int)
    for r in rows:
        big[r["Kernel_Name"].split("(")[0]] = max(big[r["Kernel_Name"].split("(")[0]], int(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if int(r["Grid_Size"]) != big[k]:
            continue                                   # only the launches with the kernel's largest grid
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("==", os.path.relpath(f, root))
    for (k, c), v in sorted(agg.items()):
        print("  %-28s %-32s mean %18.1f  n=%d  (kernel ms under PMC %.3f)" % (k, c, sum(v) / len(v), len(v), sum(dur[k]) / len(dur[k])))
