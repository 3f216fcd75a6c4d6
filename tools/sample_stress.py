#!/usr/bin/env python3
"""Stress comparison of the draw kernel against the CPU oracle on many rows (explicit noise): reports every row whose id differs and
the oracle's ratio of its two best keys there (a ratio far from 1 is a wrong draw, not a tie)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream
from oracle import mebt_oracle as orc
lib = _lib.load()
V = 16384
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
R = 4096
for scale in (1.0, 3.0):
    logits = torch.randn(R, V, generator=g) * scale
    noise = torch.empty(R, V).exponential_(generator=g)
    for temp, k in ((1.0, 0), (1.0, 32), (0.9, 32), (1.0, 5), (1.0, 256)):
        ids_r, probs_r = orc.sample_from_logits(logits, temp, k or None, None, noise)
        ids = torch.empty(R, dtype=torch.long, device="cuda"); score = torch.empty(R, device="cuda")
        probs = torch.empty(R, V, device="cuda")
        ld, nd = logits.cuda(), noise.cuda()
        check(lib.mebt_op_sample(ptr(ld), ptr(nd), temp, k, 0.0, ptr(ids), ptr(score), ptr(probs), R, V, cur_stream()))
        torch.cuda.synchronize()
        mism = (ids.cpu() != ids_r).nonzero().flatten().tolist()
        kept_bad = int(((probs.cpu() > 0) != (probs_r > 0)).any(1).sum())
        worst = 1.0
        for r in mism:
            top2 = (probs_r[r].double() / noise[r].double()).topk(2).values
            worst = max(worst, float(top2[0] / top2[1]))
        print(f"scale {scale} temp {temp} top_k {k}: {len(mism)} id mismatches (worst top-2 key ratio {worst:.6f}), rows with a different kept set: {kept_bad}", flush=True)
        if kept_bad:
            r = int(((probs.cpu() > 0) != (probs_r > 0)).any(1).nonzero()[0])
            print("   row", r, "kept hip", int((probs[r] > 0).sum()), "kept oracle", int((probs_r[r] > 0).sum()))
