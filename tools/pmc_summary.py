#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name."""
import csv, sys, collections, re
def short(n):
    m = re.match(r"(?:void )?(?:str_er::)?([A-Za-z_0-9]+)", n)     # "void str_er::k_tile_tree<480>(...)" -> k_tile_tree
    return m.group(1) if m else n
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("==", path)
    for k in sorted(acc, key=lambda k: -sum(dur[k])):
        d = dur[k]
        print(f"{k:18s} calls {len(d):4d} avg_us {sum(d)/len(d)/1e3:9.1f} lds_bytes? ", end="")
        print("  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())))
