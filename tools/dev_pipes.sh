#!/bin/bash
# Developer aid (GPU box): the headline at several numbers of batches in flight, then the share of wall time with no kernel running (tools/dev_trace2.sh)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/gpurun_out/pipes; mkdir -p $OUT
for p in ${PLIST:-6 2 3 4 8 6}; do
  echo -n "pipelines $p: " | tee -a $OUT/pipes.txt
  python bench.py --no-cpu-baseline --no-latency --no-host-frames --no-ocr-legs --no-4k-leg --no-ties-leg --pipelines $p 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_min'], d['value_max'], d['ms_per_step'], d['config']['batches_per_step'], d['config'].get('host_cpus_busy'))" | tee -a $OUT/pipes.txt
done
MODES=0 PIPES=6 bash tools/dev_trace2.sh 2>&1 | tee -a $OUT/pipes.txt
