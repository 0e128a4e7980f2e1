#!/bin/bash
# Developer aid (GPU box): memory-side counters of the config-3 scorer kernels (one batch in flight)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-ocr_pmc_mem}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --ocr --steps 4 --warmup 1 --repeats 1 --pipelines 1 --no-latency --no-host-frames --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_a -o p -- $B > $OUT/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d /tmp/pmc_b -o p -- $B > $OUT/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d /tmp/pmc_c -o p -- $B > $OUT/pmc_c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_IFETCH --output-format csv -d /tmp/pmc_d -o p -- $B > $OUT/pmc_d.log 2>&1
python - > $OUT/summary.txt <<PY
import csv, glob, collections
for d in ("/tmp/pmc_a", "/tmp/pmc_b", "/tmp/pmc_c", "/tmp/pmc_d"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "svm" in n:
                acc[n.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
cat $OUT/summary.txt
tail -3 $OUT/pmc_c.log $OUT/pmc_d.log | cut -c1-300
