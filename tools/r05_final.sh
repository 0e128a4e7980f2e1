#!/bin/bash
# round 5: the round-end checks as the driver runs them -- smoke(), the GPU suite, the default bench line (timed)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r05final}
mkdir -p $OUT
cd $ROOT
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $OUT/smoke.log 2>&1; tail -5 $OUT/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python -c "
import json; j=json.load(open('$OUT/bench.json')); print(j['value'], j['value_min'], j['value_max'], j['roofline']['frac']); print({k:(j[k].get('value'), j[k].get('frac_of_value')) for k in ('config3_ocr_leg','group_ocr_leg','config5_4k_leg','nms_ties_leg','pcie_inclusive','pcie_inclusive_nv12')}); print(j['latency_1frame']['ms_per_frame'])"
