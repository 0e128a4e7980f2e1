#!/bin/bash
# Developer aid (GPU box): tools/dev_kprof.sh TAG [bench.py args]: per-kernel times (rocprofv3 --kernel-trace, one batch in flight) of a short bench run
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-kprof}; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kp -o k -- python $ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --pipelines 1 --min-region-s 0 --no-cpu-baseline --no-latency --no-host-frames --no-ties-leg --no-ocr-legs --no-4k-leg "$@" > $OUT/run.json 2> $OUT/run.err
python $ROOT/tools/dev_kstats.py $(find /tmp/kp -name "*kernel_trace.csv" | head -1) 9 | tee $OUT/kstats.txt | head -${KPROF_LINES:-24}
