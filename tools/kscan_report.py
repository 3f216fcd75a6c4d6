#!/usr/bin/env python3
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = []
for r in csv.DictReader(open(f)):
    if 'gemm' in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])))
rows.sort()
i = 0
for nostore in (0, 1):
    for (M, N) in ((1536, 1024), (3072, 1024), (1536, 4096), (3072, 4096)):
        out = []
        for K in (64, 256, 1024, 4096):
            grp = rows[i:i + 35]; i += 35
            d = [(e - s) / 1e3 for s, e, *_ in grp[5:]]
            out.append(f"K={K}: {sum(d) / len(d):6.2f}")
        print("nostore" if nostore else "store  ", M, N, "   ".join(out))
