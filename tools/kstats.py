#!/usr/bin/env python3
"""Print name / calls / average / min / max (ns) of the first rows of a rocprofv3 --stats run.
   tools/kstats.py <output dir> [rows]"""
import csv, glob, sys
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:rows]:
        print(r["Name"].split("(")[0][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
