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
        # 12 launches, the first two dropped: minimum and median (single launches of the text-like batch scatter by +-1.5 %)
        echo -n "   $kind: "; DEV_STOP_ITERS=12 timeout 300 python $ROOT/tools/dev_stop.py $kind 2>&1 | grep "tile_tree alone" | tail -10 |
            awk '{print $(NF-1)}' | sort -n | awk '{v[NR]=$1} END {printf "min %.4f  median %.4f ms\n", v[1], (v[int((NR+1)/2)] + v[int(NR/2)+1]) / 2}'
    done
done
