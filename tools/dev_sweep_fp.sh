#!/bin/bash
# Developer aid (GPU box): frames per batch x batches in flight
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for cfg in "48 6" "64 6" "32 6" "48 4" "64 4" "96 4" "48 8" "48 6"; do
    set -- $cfg
    python $ROOT/bench.py --steps 20 --warmup 3 --frames-per-gpu $1 --pipelines $2 --no-cpu-baseline --no-latency --no-host-frames --no-ties-leg 2>/dev/null |
        python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('F=$1 P=$2', d['value'], d['ms_per_step'])"
done
