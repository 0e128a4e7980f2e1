#!/bin/bash
# Developer aid (GPU box): per-kernel times (rocprofv3 --kernel-trace) and instruction counters of the tile kernels for one pyr3x8 batch of 48
# text-like frames, with the default split between k_tile_tree and k_tile_tree2 (or STR_ER_TILE2 from the environment).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/t2prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t2k /tmp/t2p
rocprofv3 --kernel-trace --output-format csv -d /tmp/t2k -o k -- python $ROOT/tools/dev_bench.py 48 text 8 0x07 > $OUT/run.txt 2>&1
python $ROOT/tools/dev_kstats.py $(find /tmp/t2k -name "*kernel_trace.csv" | head -1) 4 | tee $OUT/kstats.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d /tmp/t2p -o p -- python $ROOT/tools/dev_bench.py 48 text 8 0x07 > /dev/null 2>&1
python - <<PY | tee $OUT/pmc.txt
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/t2p/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r.get("Kernel_Name", "")
    if "k_tile_tree" not in n: continue
    acc[n.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    w = m.get("SQ_WAVES", 1) or 1
    print(n, "per launch:", {k: round(v) for k, v in m.items()}, "per wave:", {k: round(v / w, 1) for k, v in m.items() if k != "SQ_WAVES"})
PY
