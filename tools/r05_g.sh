#!/bin/bash
# round 5: the two-stream / CU-partition variants again with 32 hardware queues (6 contexts x 4 streams = 24 > the 16 the bench asks for by default)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-r05g}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="python $ROOT/bench.py --steps 20 --no-ocr-legs --no-4k-leg --no-latency --no-host-frames --no-ties-leg --no-cpu-baseline"
export GPU_MAX_HW_QUEUES=32
timeout 200 $Q > $OUT/q32_none.json 2> $OUT/q32_none.err
STR_ER_CU_PARTITION=64 STR_ER_CU_PARTITION_MODE=full timeout 200 $Q > $OUT/q32_full.json 2> $OUT/q32_full.err
STR_ER_CU_PARTITION=64 STR_ER_CU_PARTITION_MODE=prio timeout 200 $Q > $OUT/q32_prio.json 2> $OUT/q32_prio.err
STR_ER_CU_PARTITION=32 timeout 200 $Q > $OUT/q32_spread32.json 2> $OUT/q32_spread32.err
STR_ER_CU_PARTITION=64 timeout 200 $Q > $OUT/q32_spread64.json 2> $OUT/q32_spread64.err
STR_ER_CU_PARTITION=64 STR_ER_CU_PARTITION_MODE=full timeout 200 $Q --pipelines 4 > $OUT/q32_full_p4.json 2> $OUT/q32_full_p4.err
export GPU_MAX_HW_QUEUES=16
timeout 200 $Q --pipelines 4 > $OUT/q16_none_p4.json 2> $OUT/q16_none_p4.err
for f in $OUT/*.json; do python -c "
import json,sys
try:
    j=json.load(open('$f')); print('$f'.split('/')[-1], j['value'], j['value_min'], j['value_max'], {k:v for k,v in j['gpu_ms_per_step_by_kernel_group_serial'].items() if k in ('tile_tree','group','classify')})
except Exception as e: print('$f', 'fail', e)
"; done
