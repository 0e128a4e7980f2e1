#!/bin/bash
# Developer aid (GPU box): workgroups of the per-record kernels for the largest plane (STR_ER_NODE_BLOCKS) against the tree passes' times
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/gpurun_out/nb; mkdir -p $OUT; : > $OUT/nb.txt
for nb in ${NBS:-24 48 96 192}; do
  for cfg in "12 12 3840 2160" "1 8 1920 1080" "48 8 1920 1080"; do
    set -- $cfg
    echo -n "nb $nb  $1 x $3x$4: " | tee -a $OUT/nb.txt
    STR_ER_NODE_BLOCKS=$nb DEV_W=$3 DEV_H=$4 timeout 300 python tools/dev_bench.py $1 text $2 0x07 2>&1 | grep -E "iter 3" | sed 's/.*seam=[0-9.]* //; s/nms.*//' | tee -a $OUT/nb.txt
  done
done
