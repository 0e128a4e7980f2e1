#!/bin/bash
# Developer aid (GPU box): the longest stretches without a running kernel in a bench run (EXTRA="--group --ocr", default: the exact-ties bench), and the long HIP API calls around them
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg
rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tg -o t -- python $ROOT/bench.py --no-cpu-baseline --no-latency --no-host-frames ${EXTRA:---sibling-order 0} --pipelines ${PIPES:-6} --steps ${STEPS:-80} --warmup 6 > /tmp/tg.log 2>/dev/null
tail -1 /tmp/tg.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python - <<'PY'
import csv, glob, collections
kf = glob.glob("/tmp/tg/**/*kernel_trace.csv", recursive=True)[0]
af = glob.glob("/tmp/tg/**/*hip_api_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-24:]) for r in csv.DictReader(open(kf)))
api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")) for r in csv.DictReader(open(af))]
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo, hi = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.7
gaps = []; cur_e = None; last = None
for s, e, n in ev:
    if cur_e is not None and s > cur_e and lo < s < hi: gaps.append((s - cur_e, cur_e, s, last, n))
    if cur_e is None or e > cur_e: cur_e, last = e, n
gaps.sort(reverse=True)
print("idle total %.2f ms of %.1f ms window; %d gaps" % (sum(g[0] for g in gaps) / 1e6, (hi - lo) / 1e6, len(gaps)))
for g in gaps[:12]:
    print("  gap %.3f ms  after %s  before %s" % (g[0] / 1e6, g[3], g[4]))
    for s, e, f, th in api:
        if e > g[1] and s < g[2] and (e - s) > 100000: print("      api %-28s %.3f ms thread %s  (starts %+.3f ms rel. gap start)" % (f, (e - s) / 1e6, th[-5:], (s - g[1]) / 1e6))
print("long API calls in window:")
c = collections.Counter(); d = collections.Counter()
for s, e, f, th in api:
    if lo < s < hi: c[f] += e - s; d[f] += 1
for f, v in c.most_common(14): print("   %-32s %8.2f ms in %5d calls" % (f, v / 1e6, d[f]))
PY
