#!/usr/bin/env python3
"""L2 hit rate per kernel from one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass (collected with --kernel-trace only):
hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, L2 section), summed over a kernel's launches.
usage: pmc_l2_hit.py <dir>"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
hit, miss, n = defaultdict(float), defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(f)):
    name = re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:64]
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_HIT_sum": hit[name] += v; n[name] += 1
    elif r["Counter_Name"] == "TCC_MISS_sum": miss[name] += v
print(f"{'kernel':64s} {'launches':>8s} {'L2 requests / launch':>20s} {'hit rate':>9s}")
for k in sorted(hit, key=lambda k: -(hit[k] + miss[k])):
    tot = hit[k] + miss[k]
    if tot <= 0 or n[k] < 2: continue
    print(f"{k:64s} {n[k]:8d} {tot / n[k]:20.3e} {100 * hit[k] / tot:8.1f}%")
