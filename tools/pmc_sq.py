#!/usr/bin/env python3
"""Where do the waves of each kernel spend their cycles?  One rocprofv3 --pmc pass (with --kernel-trace only) of
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES, summed per kernel (template arguments kept, so every tile / ring variant is its own row).
MI355X_MICROARCH.md: WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.
usage: pmc_sq.py <dir> [out.json]"""
import csv, glob, json, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    m = re.search(r"(gemm_bf16_\w+|gemm_pair_kernel|wgrad_grouped_kernel|attn_\w+|ln_\w+_kernel|adamw_kernel|ce_\w+_kernel)(<[^>]*>)?", name)
    if not m:
        continue
    k = m.group(1) + (m.group(2) or "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
out = {}
print(f"{'kernel':58s} {'n':>5s} {'parked%':>8s} {'stall%':>7s} {'issue%':>7s} {'lds-stall%':>10s} {'bank-conf% of LDS cyc':>22s} {'MFMA busy / wave cyc':>21s}")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    row = {"launches": len(calls[k]), "wave_cycles": w, "parked": c.get("SQ_WAIT_ANY", 0) / w, "issue_stall": c.get("SQ_WAIT_INST_ANY", 0) / w,
           "issuing": c.get("SQ_ACTIVE_INST_ANY", 0) / w, "lds_issue_stall": c.get("SQ_WAIT_INST_LDS", 0) / w,
           "lds_bank_conflict_share": c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0),
           "mfma_busy_per_wave_cycle": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / w}
    out[k] = row
    print(f"{k[:58]:58s} {row['launches']:5d} {100 * row['parked']:8.1f} {100 * row['issue_stall']:7.1f} {100 * row['issuing']:7.1f} {100 * row['lds_issue_stall']:10.1f} "
          f"{100 * row['lds_bank_conflict_share']:22.1f} {row['mfma_busy_per_wave_cycle']:21.3f}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
