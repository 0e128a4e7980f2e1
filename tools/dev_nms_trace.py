#!/usr/bin/env python3
"""Developer aid: k_nms workgroups (one per plane) traced in place (-DSTR_ER_WG_TRACE build): cycles per part of the kernel, per plane of ONE frame."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S
W, H = 1920, 1080
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, n_pyr_levels=8, channel_mask=7))
f.set_profiling(True)
f.load_cascade(0, sp); f.load_cascade(1, wp)
d = torch.from_numpy(S.synth.frames_bgr("text", 0, 1, W, H)).cuda()
L = S.load_library()
out = (C.c_ulonglong * (512 * 16))()
for it in range(3):
    L.str_er_debug_wg_trace(out, 1)
    r = f.detect_bgr_device(d.data_ptr(), W, H, 1)
L.str_er_debug_wg_trace(out, 0)
t = np.array(out[:], dtype=np.float64).reshape(512, 16)[128:256]
t = t[(t[:, 0] > 0) & (t[:, 5] > 0)]
print("nms ms", r.profile["nms"], "planes traced", len(t))
names = ["init", "level loop", "watch list", "(mark)", "chain evaluation", "ranking"]
print("plane   K  maxl  pool | " + " ".join(f"{n:>16s}" for n in ("init", "level loop", "watch", "chains", "ranking", "total")))
for row in t[np.argsort(-(t[:, 5] - t[:, 0]))][:12]:
    d_ = [row[1] - row[0], row[2] - row[1], row[3] - row[2], row[4] - row[3], row[5] - row[4], row[5] - row[0]]
    print(f"      {int(row[7]):4d} {int(row[8]):4d} {int(row[9]):5d} | " + " ".join(f"{v:16.0f}" for v in d_))
