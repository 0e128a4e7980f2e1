#!/bin/bash
# Developer aid (GPU box): sweep of k_group_merge's group size / kernel variant (STR_ER_GROUP_X / _Y / _KERNEL): tools/dev_groups.sh WORKLOAD KIND "gx gy k" ...  (EXTRA="--size 4k": other bench flags)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
wl=$1; kind=$2; shift 2
for cfg in "$@"; do
    set -- $cfg
    STR_ER_GROUP_X=$1 STR_ER_GROUP_Y=$2 STR_ER_GROUP_KERNEL=$3 python $ROOT/bench.py $EXTRA --steps 10 --warmup 3 --workload $wl --kind $kind --no-cpu-baseline --no-latency --no-host-frames --no-ties-leg --no-ocr-legs --no-4k-leg 2>/dev/null |
        python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['gpu_ms_per_step_by_kernel_group_serial']; print('$wl $kind $1x$2 k$3', d['value'], d['ms_per_step'], {k: g[k] for k in ('group','seam','resolve','accumulate')})"
done
