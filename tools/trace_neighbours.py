#!/usr/bin/env python3
"""Which kernels sit next to the runtime's copy / fill kernels in a step?  Reads a rocprofv3 --kernel-trace CSV and prints,
for every distinct (previous kernel, copy kernel, next kernel) triple, how often it occurs.  GPU box:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x/trace -o b -- python3 bench.py --steps 3 --warmup 2 --secondary none --no-cpu-baseline
  python3 tools/trace_neighbours.py gpurun_out/x/trace"""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
trip = collections.Counter()
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if "copyBuffer" in n or "fillBuffer" in n or n.startswith("void at::native"):
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        trip[(prev, short(n), nxt)] += 1
print(f"{len(rows)} kernel records")
for (p, c, n), k in trip.most_common(60):
    print(f"{k:5d}  {p}  ->  [{c}]  ->  {n}")
