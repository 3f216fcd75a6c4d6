#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 `--kernel-trace --stats` summary (kernel_stats.csv): calls, total ms, average us, share, with the
template arguments kept and the families summed at the end.  usage: kernel_table.py <kernel_stats.csv> [top N]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
fam = {}
print(f"{'kernel':96s} {'calls':>7s} {'total ms':>9s} {'avg us':>8s} {'share':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*\)$", "", name)
    print(f"{name[:96]:96s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['AverageNs']) / 1e3:8.1f} {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
for r in rows:
    n = r["Name"]
    k = ("gemm family" if re.search(r"gemm_bf16|gemm_pair|wgrad_grouped|splitk_reduce|gemm_f32", n) else "attention" if "attn_" in n else
         "layernorm / colsum" if re.search(r"ln_|colsum", n) else "vqgan conv" if "conv3d" in n else "groupnorm" if re.search(r"gn_", n) else
         "sampler" if re.search(r"sample_kernel|scatter_ids|next_mask|topk", n) else "cross-entropy" if "ce_" in n else
         "adamw" if "adamw" in n else "embed" if "embed" in n else "codebook" if "codebook" in n or "embedding_rows" in n else "other")
    c, t = fam.get(k, (0, 0.0))
    fam[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
print()
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:24s} calls {c:7d}  total {t / 1e6:9.3f} ms  {100 * t / tot:5.1f}%")
print(f"{'all kernels':24s} total {tot / 1e6:9.3f} ms")
