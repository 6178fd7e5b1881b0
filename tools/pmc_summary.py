"""Summarises rocprofv3 --pmc CSVs written by tools/pmc_passes.sh: per kernel, the sum of each counter over all dispatches
and (for the bounce kernel) the per-bounce-index breakdown of the first pass.  usage: python3 tools/pmc_summary.py <outdir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
for f in sorted(glob.glob(os.path.join(out, "pass*", "*", "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in sorted(tot):
    print("==", k)
    for c in sorted(tot[k]):
        n = len(ndisp[(k, c)])
        print("   %-32s total %.6g   per dispatch %.6g   (%d dispatches)" % (c, tot[k][c], tot[k][c] / max(n, 1), n))
