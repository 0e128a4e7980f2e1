#!/usr/bin/env python3
"""Developer aid: how much do kernels of different batches in flight overlap?  Reads a rocprofv3 --kernel-trace CSV: the union of the kernels'
[start, end) intervals against the sum of their durations, per kernel name the share of its time during which another kernel ran too."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows), key=lambda x: x[0])
# the middle of the run (steady state: no warm-up, no single-pipeline calibration at the end)
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.75)
t_lo, t_hi = ev[int(len(ev) * lo)][0], ev[int(len(ev) * hi)][0]
ev = [e for e in ev if t_lo <= e[0] < t_hi]
tot = sum(e[1] - e[0] for e in ev)
pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
busy = 0; depth = 0; last = pts[0][0]; by_depth = collections.Counter()
for t, d in pts:
    if depth > 0: busy += t - last
    by_depth[depth] += t - last
    depth += d; last = t
span = pts[-1][0] - pts[0][0]
print(f"kernels {len(ev)}, span {span/1e6:.2f} ms, sum of durations {tot/1e6:.2f} ms, union busy {busy/1e6:.2f} ms ({100*busy/span:.1f} % of the span), mean concurrency while busy {tot/busy:.2f}")
print("time by number of kernels running:", {k: f"{100*v/span:.1f}%" for k, v in sorted(by_depth.items())})
names = collections.Counter()
for s, e, n in ev: names[n] += e - s
for n, v in names.most_common(12): print(f"  {n:40s} {v/1e6:8.2f} ms  {100*v/tot:5.1f} %")

# the idle gaps (no kernel running): how long, and what ended before / started after the longest ones
gaps = []
end = None
for st, en, n in sorted(ev):
    if end is not None and st > end:
        gaps.append((st - end, last_name, n, end))
    if end is None or en > end:
        end, last_name = en, n
gaps.sort(reverse=True)
tot_gap = sum(g[0] for g in gaps)
print(f"idle gaps: {len(gaps)}, total {tot_gap/1e6:.2f} ms; >= 100 us: {sum(1 for g in gaps if g[0] >= 100e3)} ({sum(g[0] for g in gaps if g[0] >= 100e3)/1e6:.2f} ms), 20-100 us: {sum(1 for g in gaps if 20e3 <= g[0] < 100e3)} ({sum(g[0] for g in gaps if 20e3 <= g[0] < 100e3)/1e6:.2f} ms), < 20 us: {sum(1 for g in gaps if g[0] < 20e3)} ({sum(g[0] for g in gaps if g[0] < 20e3)/1e6:.2f} ms)")
for g in gaps[:12]:
    print(f"  {g[0]/1e3:8.1f} us after {g[1][:34]:34s} before {g[2][:34]}")
