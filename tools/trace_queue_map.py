import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
m = collections.defaultdict(collections.Counter)
for r in rows: m[r["Stream_Id"]][r["Queue_Id"]] += 1
for st in sorted(m): print("  stream", st, dict(m[st]))
