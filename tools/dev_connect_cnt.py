#!/usr/bin/env python3
"""Developer aid: per-wave counts of the hand-written connect loop (library built with -DSTR_ER_CONNECT_CNT: tools/dev_build_var.sh cnt -DSTR_ER_CONNECT_CNT;
STR_ER_LIB=.../lib/var/cnt.so python tools/dev_connect_cnt.py [text|noise] [channel mask])."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["STR_ER_DEBUG_TILE_ONLY"] = "1"
import torch
import str_er_amd as S
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
mask = int(sys.argv[2], 0) if len(sys.argv) > 2 else 7
F, W, H = 8, 1920, 1080
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=8, channel_mask=mask))
f.load_cascade(0, sp); f.load_cascade(1, wp)
d = torch.from_numpy(S.synth.frames_bgr(kind, 0, F, W, H)).cuda()
L = S.load_library()
out = (C.c_ulonglong * 8)()
for it in range(2):
    L.str_er_debug_connect_counts(out, 1)
    try:
        f.detect_bgr_device(d.data_ptr(), W, H, F)
    except S.StrErError:
        pass
L.str_er_debug_connect_counts(out, 0)
w = max(1, out[0])
print(f"{kind} mask {mask}: waves {out[0]}, per wave: edges {out[4] / w:.1f}, loop iterations {out[1] / w:.2f}, walk rounds a {out[2] / w:.2f} b {out[3] / w:.2f}")
