#!/usr/bin/env python3
"""Developer aid: per-phase cycle split of k_tile_tree (needs a -DSTR_ER_PHASE_PROF build)."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S
F = 8; W, H = 1920, 1080
kind = sys.argv[1] if len(sys.argv) > 1 else 'text'
tmp = tempfile.mkdtemp(); sp, wp = S.cascade_io.write_golden(tmp)
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F))
f.set_profiling(True)
f.load_cascade(0, sp); f.load_cascade(1, wp)
src = S.synth.frames_bgr(kind, 0, 4, W, H)
d = torch.from_numpy(np.stack([src[i % 4] for i in range(F)])).cuda()
L = S.load_library()
out = (C.c_ulonglong * 16)()
f.detect_bgr_device(d.data_ptr(), W, H, F)
L.str_er_debug_phase_cycles(out, 1)
r = f.detect_bgr_device(d.data_ptr(), W, H, F)
L.str_er_debug_phase_cycles(out, 1)
names = ['load+prelink', 'h-edges', 'v-edges', 'flatten', 'ids', 'run stats', 'seam-map', 'bottom-up']
tot = sum(out[i] for i in range(8))
ntiles = F * 6 * 30 * 34
print(kind, 'tile_tree ms', r.profile['tile_tree'], 'tiles', ntiles)
for i, n in enumerate(names):
    print(f'  {n:12s} {out[i]/ntiles:10.0f} ticks/tile  {100*out[i]/tot:5.1f}%')

for i, n in enumerate(['connects', 'loop iters', 'find hops', 'cas', 'cas fail']):
    print(f'  {n:12s} {out[8+i]/ntiles:10.1f} per tile')

nb = max(out[15], 1)
print(f'classify: blocks {out[15]}  phase1 {out[12]/nb/100:.1f} us  strong {out[13]/nb/100:.1f} us  weak {out[14]/nb/100:.1f} us  (classify ms {r.profile["classify"]:.3f})')
