#!/usr/bin/env python3
"""What runs between two LDPC launches? Reads a rocprofv3 --kernel-trace csv of bench.py and lists, for the last few decodes, the
gap to the previous decode and the kernels that executed (partly) inside it."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:36]) for r in csv.DictReader(open(f))]
rows.sort()
ld = [r for r in rows if "ldpc_decode" in r[2]]
for prev, cur in list(zip(ld[:-1], ld[1:]))[-4:-1]:
    gap0, gap1 = prev[1], cur[0]
    print("LDPC %.3f ms | gap %.3f ms" % ((cur[1] - cur[0]) / 1e6, (gap1 - gap0) / 1e6))
    agg = {}
    for s, e, n in rows:
        if e <= gap0 or s >= gap1 or "ldpc_decode" in n:
            continue
        a = agg.setdefault(n, [0, 0.0, 1e18, 0])
        a[0] += 1; a[1] += (min(e, gap1) - max(s, gap0)) / 1e3; a[2] = min(a[2], (max(s, gap0) - gap0) / 1e3); a[3] = max(a[3], (min(e, gap1) - gap0) / 1e3)
    for n, (c, busy, first, last) in sorted(agg.items(), key=lambda kv: kv[1][2]):
        print("   %-38s x%-4d busy %8.1f us   window %8.1f .. %8.1f us" % (n, c, busy, first, last))
