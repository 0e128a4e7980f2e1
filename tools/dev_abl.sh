#!/bin/bash
# Developer aid (GPU box): per-kernel-group times of a 4K batch, of one 1080p frame and of a 48-frame batch with the default library and with the variant
# libraries named in VARS (lib/var/NAME.so, tools/dev_build_var.sh).  Round 6 used it with timing-only builds that leave parts of k_resolve / k_reduce / k_seam
# out (results WRONG in such builds; today only -DSTR_ER_ABL_SEAM is left in the source: k_seam without its connects) -- DESIGN 3.2 "Round 6"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/gpurun_out/abl; mkdir -p $OUT; : > $OUT/abl.txt
for v in "" ${VARS:- }; do
  lib=; [ -n "$v" ] && lib=$ROOT/scene-text-recognition_amd/lib/var/$v.so
  echo "== ${v:-default} 4K x12" | tee -a $OUT/abl.txt
  STR_ER_LIB=$lib DEV_W=3840 DEV_H=2160 timeout 300 python tools/dev_bench.py 12 text 12 0x07 2>&1 | grep -E "iter [23]" | sed 's/.*cands/cands/' | tee -a $OUT/abl.txt
  echo "== ${v:-default} 1080p x1" | tee -a $OUT/abl.txt
  STR_ER_LIB=$lib timeout 300 python tools/dev_bench.py 1 text 8 0x07 2>&1 | grep -E "iter [23]" | sed 's/.*cands/cands/' | tee -a $OUT/abl.txt
  echo "== ${v:-default} 1080p x48" | tee -a $OUT/abl.txt
  STR_ER_LIB=$lib timeout 300 python tools/dev_bench.py 48 text 8 0x07 2>&1 | grep -E "iter [23]" | sed 's/.*cands/cands/' | tee -a $OUT/abl.txt
done
