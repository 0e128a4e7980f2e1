#!/bin/bash
# round 5, GPU session 3: tests, OCR legs after the coupling rewrite, CU-partition variants
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-r05c}
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py --no-4k-leg --no-host-frames --no-latency --no-ties-leg > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
Q="python $ROOT/bench.py --steps 20 --no-ocr-legs --no-4k-leg --no-latency --no-host-frames --no-ties-leg --no-cpu-baseline"
STR_ER_CU_PARTITION=64 STR_ER_CU_PARTITION_MODE=full timeout 200 $Q > $OUT/cu_full.json 2> $OUT/cu_full.err
STR_ER_CU_PARTITION=64 STR_ER_CU_PARTITION_MODE=xcd timeout 200 $Q > $OUT/cu_xcd64.json 2> $OUT/cu_xcd64.err
STR_ER_CU_PARTITION=32 STR_ER_CU_PARTITION_MODE=xcd timeout 200 $Q > $OUT/cu_xcd32.json 2> $OUT/cu_xcd32.err
timeout 200 $Q > $OUT/cu_none.json 2> $OUT/cu_none.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ocr -o s -- python $ROOT/bench.py --ocr --steps 6 --warmup 2 --repeats 1 --pipelines 1 --no-latency --no-host-frames --no-cpu-baseline > $OUT/prof_ocr.log 2>&1
rm -f $OUT/prof_ocr/*/*kernel_trace.csv $OUT/prof_ocr/*kernel_trace.csv
ls $OUT
