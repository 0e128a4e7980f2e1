#!/usr/bin/env python3
"""Developer aid: how often do NMS sibling ties occur on the synthetic frames, and how many survive the pruning?
python tools/dev_ties.py [first_frame] [n_frames] [kind] [sibling_order]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["STR_ER_DEBUG_STATS"] = "1"
import str_er_amd as S
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
order = int(sys.argv[4]) if len(sys.argv) > 4 else 2
W, H = 1920, 1080
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=16, n_pyr_levels=8, channel_mask=7, sibling_order=order))
for i0 in range(first, first + n, 16):
    frames = S.synth.frames_bgr(kind, i0, 16, W, H)
    for rep in range(2):
        sys.stderr.write(f"== frames {i0}..{i0 + 15} rep {rep}\n"); sys.stderr.flush()
        r = f.text_detect(frames, S.STAGE_EXTRACT | S.STAGE_NMS)
        sys.stderr.write(f"   ambiguous planes: {[(int(p.frame), p.ch, p.pyr, p.ambiguous) for p in r.planes if p.ambiguous]}\n")
