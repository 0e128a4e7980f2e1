#!/bin/bash
# Sanitizer builds of the HOST side of libstr_er_hip.so (SURVEY section 5 "race detection / sanitizers"):
#   tools/san_build.sh asan   -> scene-text-recognition_amd/lib/san/libstr_er_hip_asan.so   (-fsanitize=address,undefined)
#   tools/san_build.sh tsan   -> scene-text-recognition_amd/lib/san/libstr_er_hip_tsan.so   (-fsanitize=thread)
# The host translation units -- flood_order.cpp (the walk pool), er_group.cpp, gather.cpp (in-process groups, RCCL), stream_api.cpp (workers),
# str_er_api.cpp (cascade / SVM / strip-blob parsers, the batch driver) -- are compiled with the sanitizer; the kernels (*.hip) are the
# product's objects as they are.  Run python under it with tools/san_run.sh (preloads the sanitizer runtime).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)/scene-text-recognition_amd
kind=${1:-asan}
case $kind in
    asan) SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan";;
    tsan) SAN="-fsanitize=thread -shared-libsan";;
    # (ASan's allocator cannot start next to the HIP runtime on a box WITH a GPU -- "allocator is trying to allocate 0x400000 bytes", inside
    # libamdhip64's initialisation --, so what needs a device runs under this one: undefined behaviour + bounds-checked std containers)
    ubsan) SAN="-fsanitize=undefined,bounds,float-cast-overflow -D_GLIBCXX_ASSERTIONS -shared-libsan";;
    *) echo "usage: $0 asan|tsan|ubsan"; exit 2;;
esac
python $ROOT/build.py > /dev/null          # the kernels' objects
OUT=$ROOT/lib/san; mkdir -p $OUT
objs=""
for src in er_group flood_order gather str_er_api api_models api_strips api_stages stream_api; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $SAN -x hip \
        -c $ROOT/csrc/$src.cpp -o $OUT/${src}_$kind.o
    objs="$objs $OUT/${src}_$kind.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -o $OUT/libstr_er_hip_$kind.so $objs $ROOT/lib/{er_kernels,ocr_kernels,track_kernels}.o -ldl -lpthread
rm -f $objs
echo $OUT/libstr_er_hip_$kind.so
