"""Summarise rocprofv3 --pmc csv outputs: python tools/pmc_summary.py <kernel substring> <csv> [<csv> ...]"""
import collections, csv, sys
key = sys.argv[1]
for f in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        print(f"{c:34s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
