#!/usr/bin/env python3
"""Developer aid: time k_tile_tree alone (STR_ER_DEBUG_TILE_ONLY); used with -DSTR_ER_STOP_AFTER=n builds."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["STR_ER_DEBUG_TILE_ONLY"] = "1"
import torch
import str_er_amd as S
F = 32; W, H = 1920, 1080
kind = sys.argv[1] if len(sys.argv) > 1 else 'text'
tmp = tempfile.mkdtemp(); sp, wp = S.cascade_io.write_golden(tmp)
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=8, channel_mask=int(os.environ.get('DEV_STOP_MASK', '7'), 0)))
f.load_cascade(0, sp); f.load_cascade(1, wp)
src = S.synth.frames_bgr(kind, 0, 4, W, H)
d = torch.from_numpy(np.stack([src[i % 4] for i in range(F)])).cuda()
for it in range(int(os.environ.get('DEV_STOP_ITERS', '4'))):
    try:
        f.detect_bgr_device(d.data_ptr(), W, H, F)
    except S.StrErError:
        pass
