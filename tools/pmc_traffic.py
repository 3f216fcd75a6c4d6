#!/usr/bin/env python3
"""HBM traffic per launch of the GEMM kernel family from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes).  Counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled.
usage: pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [out.json]"""
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = {}, {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        k = "gemm_bf16" if ("gemm_bf16" in name or "wgrad_grouped" in name or "gemm_pair" in name) else name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
        n[k] = n.get(k, 0) + 1
    return tot, n


ft, fn = per_kernel(sys.argv[1], "FETCH_SIZE")
wt, wn = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(ft, key=lambda k: -ft[k]):        # every kernel (the embed gather / scatter rows are small but asked for)
    fetch = 2.0 * ft[k] * 1024 / fn[k]          # gfx950 correction: x2
    write = wt.get(k, 0.0) * 1024 / max(1, wn.get(k, 1))
    out[k] = {"launches": fn[k], "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": fetch + write}
    print(f"{k:60s} n={fn[k]:5d} fetch {fetch / 1e6:8.2f} MB  write {write / 1e6:8.2f} MB per launch")
if len(sys.argv) > 3:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mebt_amd.launch import csrc_fingerprint
    out["_csrc_sha256"] = csrc_fingerprint()        # bench.py nulls `roofline.traffic` when the kernels changed after this profile
    json.dump(out, open(sys.argv[3], "w"), indent=1)
