#!/bin/bash
# Developer aid (GPU box): the SVM tests on the 120-samples-per-class model, then the config-3 leg (both models) of a short bench run.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/ocr120; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_svm.py -x -q -m gpu -k "120 or reference_vectors or other_class" 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 1200 python bench.py --steps 20 --no-ties-leg --no-host-frames --no-latency --no-4k-leg --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY | tee $OUT/summary.txt
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], d["config"].get("batches_per_step"))
for k in ("config3_ocr_leg", "group_ocr_leg"):
    v = d.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "frac_of_value", "gpu_ms_per_batch_isolated", "ers_scored_per_batch", "small_model_5_per_class")})
    if "roofline_svm_kernel" in v: print("   ", {x: v["roofline_svm_kernel"].get(x) for x in ("achieved", "frac", "avg_launch_ms", "algorithmic_tflops", "frac_algorithmic")})
PY
tail -5 $OUT/bench.err
