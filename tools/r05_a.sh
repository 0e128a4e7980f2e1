#!/bin/bash
# round 5, first GPU session: baselines for config 3 (OCR scorer), config 5 (4K) and the plain path on ONE box
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline"
timeout 300 $B --steps 20 > $OUT/plain.json 2> $OUT/plain.err
timeout 300 $B --steps 20 --ocr --no-latency > $OUT/ocr.json 2> $OUT/ocr.err
timeout 300 $B --steps 20 --group --ocr --no-latency --no-host-frames > $OUT/group_ocr.json 2> $OUT/group_ocr.err
timeout 300 $B --steps 20 --group --no-latency --no-host-frames > $OUT/group.json 2> $OUT/group.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ocr -o s -- $B --ocr --steps 6 --warmup 2 --pipelines 1 --no-latency --no-host-frames > $OUT/prof_ocr.log 2>&1
rm -f $OUT/prof_ocr/*/*kernel_trace.csv $OUT/prof_ocr/*kernel_trace.csv
timeout 400 $B --size 4k --steps 12 > $OUT/k4_text.json 2> $OUT/k4_text.err
timeout 400 $B --size 4k --steps 6 --kind noise --no-host-frames > $OUT/k4_noise.json 2> $OUT/k4_noise.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_4k -o s -- $B --size 4k --steps 6 --warmup 2 --pipelines 1 --no-latency --no-host-frames --no-ties-leg > $OUT/prof_4k.log 2>&1
rm -f $OUT/prof_4k/*/*kernel_trace.csv $OUT/prof_4k/*kernel_trace.csv
ls -R $OUT | head -50
