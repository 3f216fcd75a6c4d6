#!/usr/bin/env python3
"""MFMA pipe utilisation of the GEMM / attention kernel families from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; collected with --kernel-trace only).
SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs (MI355X_MICROARCH.md: it counts cycles, 32 per 32x32x16 bf16 MFMA,
i.e. 16 per v_mfma_f32_16x16x32_bf16 — checked: 7 steps x 4.0 TFLOP / 16384 flop x 16 = 2.73e10 = the counter);
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.  utilisation = busy cycles / (GUI_ACTIVE / 8 x 1024 SIMDs).
Kernels run a little slower under counter collection, so this reads ~10 % below the event-timed roofline fraction.
usage: pmc_mfma.py <dir> [out.json]"""
import csv, glob, json, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
busy, act, n = defaultdict(float), defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if re.search(r"gemm_bf16|gemm_pair|wgrad_grouped", name): k = "gemm_family"
    elif "attn_" in name: k = re.search(r"attn_\w+", name).group(0)
    else: continue
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": busy[k] += v; n[k] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": act[k] += v
out = {}
for k in busy:
    util = busy[k] / (act[k] / 8 * 1024) if act[k] else 0.0
    out[k] = {"launches": n[k], "mfma_busy_cycles": busy[k], "gui_active_cycles": act[k], "mfma_util": util}
    print(f"{k:22s} launches {n[k]:5d}  MFMA busy {busy[k]:.3e} cyc  active {act[k]:.3e} cyc  util {100 * util:5.1f} %")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
