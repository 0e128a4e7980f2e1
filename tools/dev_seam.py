#!/usr/bin/env python3
"""Developer aid: work counters of k_seam (needs a -DSTR_ER_SEAM_PROF build)."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S
F = 8; W, H = 1920, 1080
kind = sys.argv[1] if len(sys.argv) > 1 else 'text'
tmp = tempfile.mkdtemp(); sp, wp = S.cascade_io.write_golden(tmp)
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=8, channel_mask=0x7))
f.set_profiling(True)
f.load_cascade(0, sp); f.load_cascade(1, wp)
src = S.synth.frames_bgr(kind, 0, 4, W, H)
d = torch.from_numpy(np.stack([src[i % 4] for i in range(F)])).cuda()
L = S.load_library()
out = (C.c_ulonglong * 8)()
f.detect_bgr_device(d.data_ptr(), W, H, F)
L.str_er_debug_seam_counts(out, 1)
r = f.detect_bgr_device(d.data_ptr(), W, H, F)
L.str_er_debug_seam_counts(out, 1)
names = ['connects', 'loop iters', 'find hops', 'cas', 'cas fail', 'same-level connects', 'pairs', '-']
print(kind, 'seam ms (instrumented)', r.profile['seam'])
for n, v in zip(names, out):
    print(f'  {n:22s} {v:12d}  {v / max(out[0], 1):8.2f} per connect')
