#!/bin/bash
# round 5: svm tests, then the config-3 leg with the default library and every library under lib/var (A/B on one box)
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
  timeout 200 $B > $OUT/default_$rep.json 2> $OUT/default_$rep.err
  for lib in $ROOT/scene-text-recognition_amd/lib/var/*.so; do
    [ -f "$lib" ] || continue
    STR_ER_LIB=$lib timeout 200 $B > $OUT/$(basename $lib .so)_$rep.json 2> $OUT/$(basename $lib .so)_$rep.err
  done
done
for f in $OUT/*_[12].json; do python -c "
import json; j=json.load(open('$f')); print('$f'.split('/')[-1], j['value'], j['value_min'], j['value_max'], {k:v for k,v in j['gpu_ms_per_step_by_kernel_group_serial'].items() if 'ocr' in k or 'svm' in k})"; done
