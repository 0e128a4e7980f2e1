#!/bin/bash
# Developer aid (GPU box): the GPU tests, then the S-ties legs at 1080p and 4K (host walk cost per plane)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --no-ocr-legs --no-host-frames --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('value', j['value'], j['value_min'], j['value_max']); t=j['nms_ties_leg']; print('ties', t['value'], t['frac_of_value'], t['flood_walk_ms_per_batch'], t['host_cpus_busy']); print('4k', j['config5_4k_leg']['value'], j['config5_4k_leg'].get('nms_ties'))"
