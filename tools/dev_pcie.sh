#!/bin/bash
# Developer aid (GPU box): the PCIe-inclusive legs with one and with two copy streams per upload, same box
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for n in 1 2 1 2; do
  echo "== STR_ER_UPLOAD_STREAMS=$n"
  STR_ER_UPLOAD_STREAMS=$n timeout 600 python bench.py --steps 20 --no-ties-leg --no-latency --no-ocr-legs --no-4k-leg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'])
for k in ('pcie_inclusive','pcie_inclusive_nv12'):
    v=d[k]; print(k,{x:v.get(x) for x in ('value','frac_of_value','h2d_gbs','h2d_gbs_link_alone')})"
done
