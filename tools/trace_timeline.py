#!/usr/bin/env python3
"""Timeline analysis of a `rocprofv3 --kernel-trace` CSV of bench.py: per train step, the wall time, the union
of kernel-busy time, per-queue busy time, idle gaps, and the kernels on the critical (main) queue.
Usage: python tools/trace_timeline.py <kernel_trace.csv> [--step-kernel adamw_kernel]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
# a step starts with the embedding kernel
ends = [i for i, r in enumerate(rows) if "embed_fwd" in r[3]]
print(f"{len(rows)} dispatches, {len(ends)} steps")


def short(n):
    for p in ("void ", "(anonymous namespace)::"):
        n = n.replace(p, "")
    i = n.find("(")
    n = n[:i] if i > 0 else n
    return n[:70]


# pick the last full timed step: between the second-to-last and last adamw at full size
WHICH = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a, b = ends[WHICH], ends[WHICH + 1]
step = rows[a:b]
t0, t1 = step[0][0], max(r[1] for r in step)
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us")
# union busy
iv = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"union busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us in {len(gaps)} gaps (mean {sum(g for g, _ in gaps) / max(1, len(gaps)) / 1e3:.2f} us)")
perq = defaultdict(int)
for s, e, q, _ in step:
    perq[q] += e - s
for q, v in perq.items():
    print(f"  queue {q}: busy {v / 1e3:.1f} us")
# time where exactly one queue is active vs both
ev = []
for s, e, q, _ in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
print("  concurrency histogram (us):", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
agg = defaultdict(lambda: [0, 0])
for s, e, q, n in step:
    k = (q, short(n))
    agg[k][0] += e - s
    agg[k][1] += 1
print("  top kernels (queue, name): total us, count, mean us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"   q{k[0]} {k[1]:72s} {v[0] / 1e3:8.1f} {v[1]:4d} {v[0] / v[1] / 1e3:7.2f}")
