#!/bin/bash
# Developer aid (GPU box): the 4K frames as the main timed region (per-kernel-group times of a batch on the line), optionally the GPU suite first (TESTS=1)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/gpurun_out/k4; mkdir -p $OUT
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt; fi
python bench.py --size 4k --no-cpu-baseline --no-latency --no-host-frames --no-ocr-legs --no-ties-leg 2>/dev/null | tail -1 > $OUT/bench_4k.json
python - <<PY
import json
d = json.load(open("$OUT/bench_4k.json"))
print("4k value", d["value"], d.get("value_min"), d.get("value_max"), "ms/step", d["ms_per_step"], d["config"].get("batches_per_step"))
print(d.get("gpu_ms_per_step_by_kernel_group_serial"))
PY
