#!/bin/bash
# Developer aid (GPU box): time k_tile_tree alone for every library variant in scene-text-recognition_amd/lib/stop/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for so in $ROOT/scene-text-recognition_amd/lib/stop/*.so; do
    for kind in text noise; do
        echo -n "$(basename $so) $kind: "
        STR_ER_LIB=$so python $ROOT/tools/dev_stop.py $kind 2>&1 | grep "tile_tree alone" | tail -2 | tr '\n' ' '; echo
    done
done
