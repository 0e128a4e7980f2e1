#!/bin/bash
# Developer aid: tools/dev_build_var.sh NAME [hipcc flags...] -- builds scene-text-recognition_amd/lib/var/NAME.so with er_kernels.hip compiled
# with the extra flags (the other objects are the default build's); prints the tile kernel's register use.
ROOT=$(cd "$(dirname "$0")/.." && pwd)/scene-text-recognition_amd
name=$1; shift
mkdir -p $ROOT/lib/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-result -x hip \
    -c $ROOT/csrc/er_kernels.hip "$@" -o $ROOT/lib/var/$name.o -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -A10 "Name: _ZN6str_er11k_tile_treeILi" | grep "Name\|Spill\|VGPRs:\|LDS\|Occupancy\|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '
echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lib/var/$name.so $ROOT/lib/var/$name.o \
    $ROOT/lib/{ocr_kernels,track_kernels,er_group,flood_order,gather,str_er_api,api_models,api_strips,api_stages,stream_api}.o -ldl -lpthread && rm -f $ROOT/lib/var/$name.o
