#!/bin/bash
# Developer aid (GPU box): whatever sweep was run last -- here frames per batch x batches in flight on the headline (successive runs on one box differ by a few percent: the first configuration is repeated at the end)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for cfg in ${CFGS:-48:6 32:6 24:6 16:6 40:6 32:8 24:8 16:8 32:5 48:6}; do
  set -- ${cfg%%:*} ${cfg##*:}
  echo -n "frames $1 pipelines $2: "
  python bench.py --frames-per-gpu $1 --pipelines $2 --no-cpu-baseline --no-latency --no-host-frames --no-ocr-legs --no-4k-leg --no-ties-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_min'], d['value_max'], 'ms/step', d['ms_per_step'], d['config']['batches_per_step'])"
done
