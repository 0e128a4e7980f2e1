#!/bin/bash
# Developer aid: where do k_tile_tree's time and VALU instructions go?  Builds the library once per phase with
# -DSTR_ER_STOP_AFTER=n (the kernel returns after phase n) into gpurun_out/stoplibs/, then (on the GPU box: `tools/dev_stop_all.sh run`)
# times the kernel alone for every variant and counts its vector / scalar / LDS / branch instructions with rocprofv3.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/scene-text-recognition_amd/lib/stop
PHASES="0 1 3 4 5 7 13 6"
# EXTRA="-DSTR_ER_TILE_V2" tools/dev_stop_all.sh build: the same for another build of the kernel
if [ "${1:-build}" = "build" ]; then
    mkdir -p $OUT; rm -f $OUT/*
    for n in $PHASES; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DSTR_ER_STOP_AFTER=$n ${EXTRA:-} -x hip \
            -c $ROOT/scene-text-recognition_amd/csrc/er_kernels.hip -o $OUT/er_kernels_$n.o &
    done
    wait
    for n in $PHASES; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libstop_$n.so $OUT/er_kernels_$n.o $ROOT/scene-text-recognition_amd/lib/{ocr_kernels,track_kernels,er_group,flood_order,gather,str_er_api,api_models,api_strips,api_stages,stream_api}.o -ldl -lpthread
        rm -f $OUT/er_kernels_$n.o
    done
    ls -la $OUT
else
    cd /tmp && export TMPDIR=/tmp
    for n in $PHASES; do
        echo "== stop after phase $n"
        STR_ER_LIB=$OUT/libstop_$n.so python $ROOT/tools/dev_stop.py ${KIND:-text} 2>&1 | grep "tile_tree alone" | tail -1
        STR_ER_LIB=$OUT/libstop_$n.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d /tmp/pmc_$n -o p -- python $ROOT/tools/dev_stop.py ${KIND:-text} > /dev/null 2>&1
        python - <<PY
import csv, glob, os
rows = []
for f in glob.glob("/tmp/pmc_$n/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = {}
for r in rows:
    if "k_tile_tree" not in r.get("Kernel_Name", ""): continue
    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
w = float(os.environ.get("DEV_STOP_WAVES", "795648"))          # waves of one launch of dev_stop.py (32 frames of pyr3x8, 3 channels)
print("   per wave:", {k: round(sum(v) / len(v) / w, 1) for k, v in acc.items()})
PY
    done
fi
