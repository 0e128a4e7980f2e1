#!/bin/bash
# round 5, GPU session 4: tests, OCR legs, stream-priority variant of the two-stream structure
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-r05d}
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py --no-4k-leg --no-host-frames --no-latency --no-ties-leg --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
Q="python $ROOT/bench.py --steps 20 --no-ocr-legs --no-4k-leg --no-latency --no-host-frames --no-ties-leg --no-cpu-baseline"
STR_ER_CU_PARTITION=64 STR_ER_CU_PARTITION_MODE=prio timeout 200 $Q > $OUT/cu_prio.json 2> $OUT/cu_prio.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ocr -o s -- python $ROOT/bench.py --ocr --steps 6 --warmup 2 --repeats 1 --pipelines 1 --no-latency --no-host-frames --no-cpu-baseline > $OUT/prof_ocr.log 2>&1
rm -f $OUT/prof_ocr/*/*kernel_trace.csv $OUT/prof_ocr/*kernel_trace.csv
ls $OUT
