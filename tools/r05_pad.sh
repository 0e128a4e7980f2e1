#!/bin/bash
# round 5: the headline line with fewer tile workgroups resident per CU (STR_ER_TILE_LDS_PAD), A/B on one box
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 20 --no-latency --no-host-frames --no-cpu-baseline --no-ocr-legs --no-ties-leg --no-4k-leg"
for rep in 1 2; do
for pad in 0 14336 6144 28672; do
  STR_ER_TILE_LDS_PAD=$pad timeout 200 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pad $pad', j['value'], j['value_min'], j['value_max'], j['roofline']['avg_launch_ms'])"
done
done
