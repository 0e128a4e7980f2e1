#!/bin/bash
# round 5: the whole GPU suite, then one run of the config-3 leg
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r05full}
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 200 python $ROOT/bench.py --ocr --steps 20 --no-latency --no-host-frames --no-cpu-baseline > $OUT/ocr.json 2> $OUT/ocr.err
python -c "
import json; j=json.load(open('$OUT/ocr.json')); print(j['value'], j['value_min'], j['value_max'], {k:v for k,v in j['gpu_ms_per_step_by_kernel_group_serial'].items() if 'ocr' in k or 'svm' in k})"
