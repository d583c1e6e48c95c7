#!/usr/bin/env python3
"""Print name / calls / average / min / max (ns) of rows of a rocprofv3 --stats run.
   tools/kstats.py <output dir> [rows | name filter]"""
import csv, glob, sys
arg = sys.argv[2] if len(sys.argv) > 2 else "4"
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows = rows[:int(arg)] if arg.isdigit() else [r for r in rows if arg in r["Name"]]
    for r in rows:
        print(r["Name"].split("(")[0][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
