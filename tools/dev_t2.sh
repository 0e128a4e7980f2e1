#!/bin/bash
# Developer aid (GPU box): the two tile kernels -- parity of the split (STR_ER_TILE2 = 0 / 1 / 2), then the tile-tree time of one pyr3x8 batch of
# 48 text-like frames with the first kernel only, the default split (chroma planes to k_tile_tree2) and everything through k_tile_tree2.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/t2; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "both_tile_kernels or bgr_frame_all_six or pyramid_planes or full_hd_frame_two" 2>&1 | tail -15 | tee $OUT/pytest.txt
for m in 0 1 2; do
    echo "== STR_ER_TILE2=$m" | tee -a $OUT/bench.txt
    STR_ER_TILE2=$m timeout 300 python tools/dev_bench.py 48 text 8 0x07 2>&1 | grep -E "iter [23]|tile2" | tee -a $OUT/bench.txt
done
