#!/usr/bin/env python3
"""Per-mode counters of tools/phase_probe.hip from rocprofv3 --pmc passes (each collected with --kernel-trace only): the probe
launches 10 kernels per mode (7 modes, in order), so dispatch k belongs to mode k // 10.  Prints, per mode, the mean of every
counter found and the derived average L1->L2 read latency (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ) and L2 hit rate.
usage: pmc_phase_probe.py <dir> [<dir> ...]"""
import csv, glob, sys
from collections import defaultdict
names = ["fill only", "stream only", "all in phase", "mixed inside each CU", "CU parity", "XCD parity", "SE parity"]
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "probe" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        order = {d_: i for i, d_ in enumerate(ids)}
        for r in rows:
            acc[order[int(r["Dispatch_Id"])] // 10][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for m in acc.values() for c in m})
print("mode".ljust(24) + "".join(c[:26].rjust(28) for c in ctrs) + "  derived")
for m in sorted(acc):
    mean = {c: sum(v) / len(v) for c, v in acc[m].items()}
    der = []
    if mean.get("TCP_TCC_READ_REQ_sum"):
        der.append(f"L1->L2 read latency {mean.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / mean['TCP_TCC_READ_REQ_sum']:.0f} clk")
    if "TCC_HIT_sum" in mean:
        der.append(f"L2 hit {100 * mean['TCC_HIT_sum'] / max(1.0, mean['TCC_HIT_sum'] + mean.get('TCC_MISS_sum', 0)):.1f} %")
    if mean.get("TCC_EA0_RDREQ_sum"):
        der.append(f"fabric read latency {mean.get('TCC_EA0_RDREQ_LEVEL_sum', 0) / mean['TCC_EA0_RDREQ_sum']:.0f} clk")
    print(names[m].ljust(24) + "".join(f"{mean.get(c, 0):28.4g}" for c in ctrs) + "  " + "; ".join(der))
