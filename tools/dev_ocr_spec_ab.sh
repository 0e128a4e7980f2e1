#!/bin/bash
# Developer aid (GPU box): the config-3 line (bench.py --ocr) with the scorer enqueued behind classify (default) and sized after the counters were read
# (STR_ER_OCR_SPEC=0), interleaved on one box; 6 and 1 batches in flight.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/ocrspec; mkdir -p $OUT
cd $ROOT
F="--ocr --steps 20 --no-ties-leg --no-host-frames --no-4k-leg --no-cpu-baseline --no-latency --no-ocr-legs"
for rep in 1 2; do
  for spec in 1 0; do
    for pl in 6 1; do
      STR_ER_OCR_SPEC=$spec timeout 600 python bench.py $F --pipelines $pl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('spec=$spec pipelines=$pl value', d['value'], 'min', d.get('value_min'), 'max', d.get('value_max'), 'ms/step', d['ms_per_step'])"
    done
  done
done | tee $OUT/ab.txt
