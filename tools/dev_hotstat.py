#!/usr/bin/env python3
"""Developer aid (GPU box; library built with -DSTR_ER_HOTSTAT): how many pieces hand over to one survivor (k_resolve) and how many children push into one parent
(k_reduce) -- the hot records of the tree passes.  Usage: tools/dev_build_var.sh hotstat -DSTR_ER_HOTSTAT; STR_ER_LIB=.../lib/var/hotstat.so python tools/dev_hotstat.py F W H levels"""
import os, sys, ctypes, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S

F, W, H, L = (int(a) for a in sys.argv[1:5])
tmp = tempfile.mkdtemp()
sp, wp = S.cascade_io.write_golden(tmp)
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=L, channel_mask=0x07))
f.load_cascade(0, sp); f.load_cascade(1, wp)
src = S.synth.frames_bgr('text', 0, min(F, 4), W, H)
d = torch.from_numpy(np.stack([src[i % len(src)] for i in range(F)])).cuda()
lib = ctypes.CDLL(os.environ["STR_ER_LIB"])
out = (ctypes.c_uint32 * 16)()
for it in range(3):
    r = f.detect_bgr_device(d.data_ptr(), W, H, F)
    lib.str_er_debug_hotstat(out)
v = list(out)
print(f"{F} x {W}x{H} x {L} levels: records {v[11]}")
print(f"  hand-overs: {v[1]} pieces to {v[9]} survivors, max {v[0]} to one; survivors with >=16 pieces: {v[2]} (they take {v[12]} pieces), >=128: {v[3]}")
print(f"  pushes: {v[5]} children into {v[10]} parents, max {v[4]} into one; parents with exactly one: {v[14]}, with >=16 children: {v[6]} (they take {v[13]} children), >=128: {v[7]}, >=1024: {v[8]}")
