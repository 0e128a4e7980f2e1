#!/usr/bin/env python3
"""Developer aid: per-kernel calls / average / total from a rocprofv3 --kernel-trace CSV, per batch (argv[2] = batches in the trace)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nb = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("str_er::", "").replace("void ", "")[:44]
    acc[n][0] += 1; acc[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in acc.values())
print(f"{'kernel':44s} {'calls/batch':>11s} {'avg us':>9s} {'ms/batch':>9s}")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:44s} {c / nb:11.2f} {t / c / 1e3:9.1f} {t / nb / 1e6:9.4f}")
print(f"{'total':44s} {'':11s} {'':9s} {tot / nb / 1e6:9.4f}")
