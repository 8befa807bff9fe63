#!/usr/bin/env python3
"""Average each PMC counter per kernel from rocprofv3 --pmc ... --output-format csv:  tools/pmc_summary.py dir [dir...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "flanhip" not in k: continue
            print(k)
            print("   " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
