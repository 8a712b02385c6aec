#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (gpurun_out/<round>_pmc_*/pmc_counter_collection.csv):
mean counter value per dispatch for each kernel family."""
import csv
import glob
import sys
from collections import defaultdict

prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r1_pmc_"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(prefix + "*/pmc_counter_collection.csv")):
    per = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = ("k_thorough" if "k_thorough" in k else "k_preplace_pairs" if "k_preplace_pairs" in k else
             "k_preplace_sites" if "k_preplace_sites" in k else
             "k_preplace" if "k_preplace" in k else "k_select" if "k_select" in k else k[:30])
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        per[(k, r["Dispatch_Id"])]["_vgpr"] = float(r.get("VGPR_Count", 0) or 0)
        per[(k, r["Dispatch_Id"])]["_lds"] = float(r.get("LDS_Block_Size", 0) or 0)
    for (k, _), d in per.items():
        for c, v in d.items():
            acc[k][c].append(v)
for k in sorted(acc):
    print("==", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("  %-24s mean/dispatch %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
