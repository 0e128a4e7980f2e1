#!/bin/bash
# Developer aid (GPU box): the PCIe-inclusive legs against the number of copy streams per upload and the batches in flight
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/gpurun_out/pcie2; mkdir -p $OUT; : > $OUT/pcie.txt
for cfg in ${CFGS:-"2 6" "3 6" "4 6" "1 6" "3 4" "4 4" "2 6"}; do
  set -- $cfg
  echo -n "upload streams $1, pipelines $2: " | tee -a $OUT/pcie.txt
  STR_ER_UPLOAD_STREAMS=$1 python bench.py --no-cpu-baseline --no-latency --no-ocr-legs --no-4k-leg --no-ties-leg --pipelines $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['pcie_inclusive']; q=d['pcie_inclusive_nv12']
print('value', d['value'], 'bgr', p['value'], p['h2d_gbs'], 'link', p['h2d_gbs_link_alone'], 'nv12', q['value'])" | tee -a $OUT/pcie.txt
done
