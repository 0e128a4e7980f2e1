#!/bin/bash
# round 5: where one frame's 0.77 ms goes -- per-stage GPU times and the kernel timeline of the 1-frame call
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r05lat}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/dev_latency.py > $OUT/latency.txt 2>&1
rm -rf /tmp/tl
DEV_LAT_N=12 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $ROOT/tools/dev_latency.py > /dev/null 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("str_er::", "").replace("void ", "")[:28], r.get("Queue_Id", "?")) for r in rows)
# last complete frame: from the last k_bgr_to_ycrcb to the end
starts = [i for i, e in enumerate(ev) if "bgr_to_ycrcb" in e[2]]
a = starts[-2]; b = starts[-1]
t0 = ev[a][0]; last = None; busy = 0
for s, e, n, q in ev[a:b]:
    gap = (s - last) / 1e3 if last else 0
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  q{q} {n}")
    busy += e - s
    last = max(last or 0, e)
print("frame span %.1f us, kernels busy %.1f us, launches %d" % ((last - t0) / 1e3, busy / 1e3, b - a))
PY
cat $OUT/latency.txt | tail -3; tail -3 $OUT/timeline.txt
