#!/bin/bash
# round 5, GPU session: tests + OCR legs + kernel stats of the config-3 run
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-r05e}
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py --no-4k-leg --no-host-frames --no-latency --no-ties-leg --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ocr -o s -- python $ROOT/bench.py --ocr --steps 6 --warmup 2 --repeats 1 --pipelines 1 --no-latency --no-host-frames --no-cpu-baseline > $OUT/prof_ocr.log 2>&1
rm -f $OUT/prof_ocr/*/*kernel_trace.csv $OUT/prof_ocr/*kernel_trace.csv
ls $OUT
