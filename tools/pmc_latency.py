#!/usr/bin/env python3
"""Average L1 -> L2 read latency and outstanding reads per CU of every kernel of the train step, from one rocprofv3 pass
`--kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum` (collected without any other trace domain):
    latency  = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ                      (clocks a 128-byte read request is outstanding)
    in flight per CU = TCP_TCC_READ_REQ_LATENCY / (kernel duration x shader clock x 256 CUs)     (Little's law)
The second figure is what profiles/r06_wgrad_overlap_evidence.txt found to be capped at ~120: a kernel near the cap is bound by
request slot-time (requests x latency), not by bytes.  usage: pmc_latency.py <dir> [shader GHz, default 2.4]"""
import csv, glob, re, sys
from collections import defaultdict
d = sys.argv[1]
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = defaultdict(lambda: defaultdict(float))
ids = defaultdict(set)
for r in csv.DictReader(open(cc)):
    name = re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    ids[name].add(r["Dispatch_Id"])
rows = []
for k, c in acc.items():
    req, lat = c.get("TCP_TCC_READ_REQ_sum", 0.0), c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0)
    ns = sum(dur.get(i, 0) for i in ids[k])
    if req <= 0 or ns <= 0 or len(ids[k]) < 2:
        continue
    rows.append((ns, k, len(ids[k]), req / len(ids[k]), lat / req, lat / (ns * ghz * 256), ns / len(ids[k]) / 1e3))
print(f"{'kernel':60s} {'launches':>8s} {'us/launch':>10s} {'read req / launch':>18s} {'latency (clk)':>14s} {'in flight per CU':>17s}")
for ns, k, n, rq, la, fl, us in sorted(rows, reverse=True):
    print(f"{k:60s} {n:8d} {us:10.1f} {rq:18.3e} {la:14.0f} {fl:17.1f}")
