"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv: per predecessor -> successor pair, mean gap in us."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rn::", "").split("<")[0]
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    if not (short(a).startswith("k_") and short(b).startswith("k_")):
        continue  # setup / torch kernels
    gaps[(short(a), short(b))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in gaps.values())
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
print("kernel time %.0f us, gaps %.0f us" % (busy, tot))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-22s -> %-22s n=%4d mean gap %6.2f us  total %8.1f" % (k[0], k[1], len(v), sum(v) / len(v), sum(v)))
