#!/usr/bin/env python3
"""Print calls / average duration (us) per kernel of a rocprofv3 --stats kernel_stats.csv.  usage: kstats.py <dir> [filter ...]"""
import csv, glob, sys
flt = sys.argv[2:]
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if flt and not any(x in row["Name"] for x in flt):
            continue
        print(f"   {row['Name'][:70]:70s} calls {row['Calls']:>7s}  avg {float(row['AverageNs']) / 1e3:8.2f} us")
