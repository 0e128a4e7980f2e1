#!/bin/bash
# round 5: A/B of the decision-value loops' unrolling (default library vs lib/var/libocr_u2.so) + tests
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r05ab}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_svm.py tests/test_track.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --ocr --steps 20 --no-latency --no-host-frames --no-cpu-baseline"
for rep in 1 2; do
  timeout 200 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err
  STR_ER_LIB=$ROOT/scene-text-recognition_amd/lib/var/libocr_u2.so timeout 200 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err
done
for f in $OUT/a_1 $OUT/b_1 $OUT/a_2 $OUT/b_2; do python -c "
import json; j=json.load(open('$f.json')); print('$f'.split('/')[-1], j['value'], j['value_min'], j['value_max'], {k:v for k,v in j['gpu_ms_per_step_by_kernel_group_serial'].items() if 'ocr' in k or 'svm' in k})"; done
