#!/bin/bash
# GPU box: the sanitizer builds against a real device -- the GPU tests that drive the concurrent host code and the byte parsers, then the fuzz loop.
# Logs -> gpurun_out/san/.  (tools/san_build.sh asan / tsan must have been run before: the .so files travel with the snapshot.)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/san; mkdir -p $OUT; rm -f $OUT/*
cd $ROOT
T="tests/test_svm.py tests/test_stream.py tests/test_gather_native.py tests/test_config5_strips.py tests/test_bench_workload.py::test_bench_workload_ties tests/test_gpu_parity.py::test_errors tests/test_track.py"
for kind in ubsan tsan; do
    echo "== $kind: pytest"
    SAN_LOG=$OUT/$kind timeout 900 bash tools/san_run.sh $kind python -m pytest $T -q -m gpu -x --deselect tests/test_config5_strips.py::test_rccl_device_blobs_world_of_one 2>&1 | tail -5 | tee $OUT/${kind}_pytest.txt
done
echo "== ubsan: fuzz"
SAN_LOG=$OUT/ubsan_fuzz timeout 300 bash tools/san_run.sh ubsan python tools/san_fuzz.py ${FUZZ_SECONDS:-60} 2>&1 | tail -5 | tee $OUT/ubsan_fuzz.txt
ls -la $OUT
for f in $OUT/ubsan.* $OUT/tsan.* $OUT/ubsan_fuzz.*; do [ -f "$f" ] && { echo "---- $f"; grep -E "ERROR|WARNING|SUMMARY|runtime error" $f | sort | uniq -c | head -20; }; done
