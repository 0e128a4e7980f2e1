#!/usr/bin/env python3
"""Developer aid: a window of a rocprofv3 --kernel-trace CSV as a timeline (start, duration, queue, kernel), with the GPU-idle gaps marked."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("str_er::", "").replace("void ", "")[:18],
              r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
t0 = ev[len(ev) * 2 // 5][0]
print("queues", collections.Counter(e[3] for e in ev).most_common(12))
last_end = 0
for s, e, n, q, st in ev:
    if s < t0 or s > t0 + win_ms * 1e6:
        continue
    gap = s - last_end if last_end and s > last_end else 0
    mark = "   <-- GPU idle %.0f us before" % (gap / 1e3) if gap > 20e3 else ""
    print(f"{(s - t0) / 1e3:9.1f} us +{(e - s) / 1e3:8.1f}  q{q:>3} s{st:>3} {n}{mark}")
    last_end = max(last_end, e)
