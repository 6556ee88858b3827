"""Diagnostic: per-UNet-call kernel table = (stats with N_CALLS=b) - (stats with N_CALLS=a), divided by (b-a)."""
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"]] = (int(r["Calls"]), int(r["TotalDurationNs"]))
    return d
a, b, n = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0))
    if c - c0 > 0:
        rows.append(((t - t0) / n / 1e6, (c - c0) / n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total ms per call %.3f" % tot)
for ms, c, k in rows[:45]:
    print("%8.3f ms %7.1f calls  %s" % (ms, c, k[:110]))
