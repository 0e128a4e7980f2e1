#!/bin/bash
# Developer aid (GPU box): counters of the config-3 scorer kernels (one batch in flight), two counter passes
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-ocr_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --ocr --steps 4 --warmup 1 --repeats 1 --pipelines 1 --min-region-s 0 --no-latency --no-host-frames --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_a -o p -- $B > $OUT/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d /tmp/pmc_b -o p -- $B > $OUT/pmc_b.log 2>&1
python - > $OUT/summary.txt <<PY
import csv, glob, collections
for d in ("/tmp/pmc_a", "/tmp/pmc_b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "ocr" in n or "svm" in n or "tile_tree" in n:
                acc[n.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
cat $OUT/summary.txt
