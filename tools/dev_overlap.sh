#!/bin/bash
# Developer aid (GPU box): how the kernels of the six batches in flight overlap (tools/dev_overlap.py on a rocprofv3 kernel trace of the headline), for the batch sizes in FS
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/overlap; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in ${FS:-48 32}; do
  rm -rf /tmp/ov_$f
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ov_$f -o t -- python $ROOT/bench.py --frames-per-gpu $f --steps 30 --warmup 6 --repeats 1 --min-region-s 0 --no-cpu-baseline --no-latency --no-host-frames --no-ocr-legs --no-4k-leg --no-ties-leg > /tmp/ov_$f.log 2>/dev/null
  echo "== frames per batch $f: $(tail -1 /tmp/ov_$f.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/overlap.txt
  python $ROOT/tools/dev_overlap.py $(find /tmp/ov_$f -name "*kernel_trace.csv" | head -1) 0.3 0.8 | tee -a $OUT/overlap.txt
done
