#!/bin/bash
# Developer aid (GPU box): run dev_cliff_one.py for the default library and every variant in scene-text-recognition_amd/lib/var/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python $ROOT/tools/dev_cliff_one.py 2>&1 | tail -1
for so in $ROOT/scene-text-recognition_amd/lib/var/*.so; do
    STR_ER_LIB=$so timeout 300 python $ROOT/tools/dev_cliff_one.py 2>&1 | tail -1
done
