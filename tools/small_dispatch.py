"""Diagnostic: total device time of small-grid dispatches in a rocprofv3 kernel trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
calls = int(sys.argv[2])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    nwg = grid // max(wg, 1)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    b = "<=8 WG" if nwg <= 8 else "<=64 WG" if nwg <= 64 else "<=256 WG" if nwg <= 256 else ">256 WG"
    k = (r["Kernel_Name"][:40], b)
    agg[k][0] += 1; agg[k][1] += d
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-42s %-9s n/call %6.1f  us/call %8.1f  avg us %6.1f" % (k[0], k[1], n / calls, us / calls, us / n))
