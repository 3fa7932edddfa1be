#!/usr/bin/env python3
"""Which hardware queue the kernels of each stream of a rocprofv3 kernel trace ran on (t_kernel_trace.csv), with the kernels that
identify the stream: the queues a process's streams get depend on the order it made them in, and a latency-bound chain that shares a
queue -- or only a pipe, queue index mod 4 -- with a queue holding a decode of milliseconds is 20 % and more slower
(profiles/HISTORY.md, round 6)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
m = collections.defaultdict(collections.Counter)
names = collections.defaultdict(collections.Counter)
for r in rows:
    m[r["Stream_Id"]][r["Queue_Id"]] += 1
    names[r["Stream_Id"]][r["Kernel_Name"].split("(")[0].replace("void ", "").replace("t2gpu::", "")[:36]] += 1
for st in sorted(m, key=int):
    top = ", ".join("%s x%d" % kv for kv in names[st].most_common(3))
    print("  stream %s queues %s: %s" % (st, dict(m[st]), top))
