#!/bin/bash
# round 5: validation after the source split + kernel stats of the --group --ocr path
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-r05h}
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_group -o s -- python $ROOT/bench.py --group --ocr --steps 6 --warmup 2 --repeats 1 --pipelines 1 --no-latency --no-host-frames --no-cpu-baseline > $OUT/prof_group.log 2>&1
rm -f $OUT/prof_group/*/*kernel_trace.csv $OUT/prof_group/*kernel_trace.csv
python - $OUT/prof_group/s_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg_ms {float(r["AverageNs"])/1e6:8.4f} pct {r["Percentage"]}')
PY
