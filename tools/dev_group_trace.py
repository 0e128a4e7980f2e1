#!/usr/bin/env python3
"""Developer aid: k_group_merge workgroups traced in place (-DSTR_ER_WG_TRACE build): cycles per part of the kernel, 48 text frames."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S
F, W, H = 48, 1920, 1080
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=8, channel_mask=7))
f.set_profiling(True)
f.load_cascade(0, sp); f.load_cascade(1, wp)
d = torch.from_numpy(S.synth.frames_bgr("text", 0, F, W, H)).cuda()
L = S.load_library()
out = (C.c_ulonglong * (512 * 16))()
for it in range(3):
    L.str_er_debug_wg_trace(out, 1)
    r = f.detect_bgr_device(d.data_ptr(), W, H, F)
L.str_er_debug_wg_trace(out, 0)
t = np.array(out[:], dtype=np.float64).reshape(512, 16)[288:384]
t = t[(t[:, 1] > 0) & (t[:, 6] > 0)]
print("group ms", r.profile["group"], "sampled workgroups", len(t))
life = t[:, 6] - t[:, 1]
print(f"lifetime mean {life.mean():.0f} ticks")
for i, n in enumerate(["load records", "connect inner seams", "hand over / canonical parents", "level loop (fold)", "store records"]):
    dt = t[:, i + 2] - t[:, i + 1]
    print(f"  {n:32s} {dt.mean():9.0f} ticks {100 * dt.mean() / life.mean():5.1f} %")
