#!/bin/bash
# Developer aid (GPU box): tools/dev_pmc.sh LIB KIND COUNTER... -- counters of k_tile_tree alone (per launch and per wave)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
lib=$1; kind=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_x
[ "$lib" != default ] && export STR_ER_LIB=$lib
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_x -o p -- python $ROOT/tools/dev_stop.py $kind > /dev/null 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("/tmp/pmc_x/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = {}
for r in rows:
    if "k_tile_tree" not in r.get("Kernel_Name", ""): continue
    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("$lib $kind per launch:", {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
