#!/bin/bash
# Developer aid (GPU box): A/B of k_tile_tree variants built with tools/dev_build_var.sh.  For every library: time of the tile kernel alone on
# text / noise frames; parity (dev_cliff_one.py) only for the names listed in $PARITY (default: the default library).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PARITY=${PARITY:-default}
for so in "" $ROOT/scene-text-recognition_amd/lib/var/*.so; do
    name=default; [ -n "$so" ] && { export STR_ER_LIB=$so; name=$(basename $so .so); }
    echo "== $name"
    case " $PARITY " in *" $name "*) timeout 300 python $ROOT/tools/dev_cliff_one.py 2>&1 | tail -1;; esac
    for kind in text noise; do
        echo -n "   $kind: "; timeout 300 python $ROOT/tools/dev_stop.py $kind 2>&1 | grep "tile_tree alone" | tail -2 | tr '\n' ' '; echo
    done
done
