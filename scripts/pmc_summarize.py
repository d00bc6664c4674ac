"""Average a rocprofv3 --pmc counter CSV per kernel name.  usage: pmc_summarize.py <dir> [...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, vals in cs.items():
                print(f"{d}\t{k[:60]}\t{c}\tn={len(vals)}\tmean={sum(vals)/len(vals):.1f}\tmax={max(vals):.1f}")
