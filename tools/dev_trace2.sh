#!/bin/bash
# Developer aid (GPU box): kernel time per step and the share of wall time with no kernel running, with exact sibling ties (0)
# and with the key rule (2)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for so in ${MODES:-0 2}; do
    rm -rf /tmp/tr_$so
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$so -o t -- python $ROOT/bench.py --no-cpu-baseline --no-latency --no-host-frames --sibling-order $so --pipelines ${PIPES:-6} --steps 20 --warmup 5 > /tmp/tr_$so.log 2>/dev/null
    tail -1 /tmp/tr_$so.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sibling $so', d['value'], d['ms_per_step'])"
    python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/tr_$so/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady state: the middle 60 % of the trace
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo, hi = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.9
busy = 0; cur_s = cur_e = None
per = collections.Counter()
for s, e, n in ev:
    if e < lo or s > hi: continue
    s = max(s, lo); e = min(e, hi)
    per[n.split("(")[0][-28:]] += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("  window %.1f ms, some kernel running %.1f %%" % ((hi - lo) / 1e6, 100 * busy / (hi - lo)))
print("  kernel time / window:", {k: round(v / (hi - lo), 3) for k, v in per.most_common(12)})
PY
done
