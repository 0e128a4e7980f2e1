#!/bin/bash
# Developer aid (GPU box): str_er_set_batch_slots against frames per batch and batches in flight (CFGS = frames:pipelines:slots ...)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for cfg in ${CFGS:-32:6:0 32:6:3 32:6:4 32:8:4 48:6:0 48:6:3 48:6:4 48:8:4 32:6:0}; do
  IFS=: read f p k <<< "$cfg"
  echo -n "frames $f pipelines $p slots $k: "
  python bench.py --frames-per-gpu $f --pipelines $p --batch-slots $k ${EXTRA:-} --no-cpu-baseline --no-latency --no-host-frames --no-ocr-legs --no-4k-leg --no-ties-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_min'], d['value_max'], 'ms/step', d['ms_per_step'], d['config']['batches_per_step'])"
done
