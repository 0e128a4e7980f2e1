#!/bin/bash
# Developer aid (GPU box): for the default library and every variant in scene-text-recognition_amd/lib/var/: parity count (dev_cliff_one.py)
# and the time of k_tile_tree alone on text / noise frames (dev_stop.py).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for so in "" $ROOT/scene-text-recognition_amd/lib/var/*.so; do
    [ -n "$so" ] && export STR_ER_LIB=$so
    echo "== ${so:-default}"
    timeout 300 python $ROOT/tools/dev_cliff_one.py 2>&1 | tail -1
    for kind in text noise; do
        echo -n "   $kind: "; timeout 300 python $ROOT/tools/dev_stop.py $kind 2>&1 | grep "tile_tree alone" | tail -2 | tr '\n' ' '; echo
    done
done
