#!/usr/bin/env python3
"""Developer aid: where does a workgroup of k_tile_tree spend its LIFETIME?  (library built with -DSTR_ER_WG_TRACE: tools/dev_build_var.sh wgtrace -DSTR_ER_WG_TRACE;
STR_ER_LIB=.../lib/var/wgtrace.so python tools/dev_wg_trace.py [text|noise] [channel mask])."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["STR_ER_DEBUG_TILE_ONLY"] = "1"
import torch
import str_er_amd as S
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
mask = int(sys.argv[2], 0) if len(sys.argv) > 2 else 7
F, W, H = 32, 1920, 1080
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=8, channel_mask=mask))
f.load_cascade(0, sp); f.load_cascade(1, wp)
src = S.synth.frames_bgr(kind, 0, 4, W, H)
d = torch.from_numpy(np.stack([src[i % 4] for i in range(F)])).cuda()
L = S.load_library()
out = (C.c_ulonglong * (512 * 16))()
for it in range(3):
    L.str_er_debug_wg_trace(out, 1)
    try:
        f.detect_bgr_device(d.data_ptr(), W, H, F)
    except S.StrErError:
        pass
L.str_er_debug_wg_trace(out, 0)
t = np.array(out[:], dtype=np.float64).reshape(512, 16)
t = t[(t[:, 15] > 0) & (t[:, 6] > 0)]
order = [0, 1, 3, 4, 5, 7, 13, 6]        # the phases in program order (PHASE_MARK ids)
names = ["load", "edges + connects", "flatten", "ids", "statistics", "fold", "export", "seam map"]
prev = t[:, 15]
life = t[:, 6] - t[:, 15]
print(f"{kind} mask {mask}: {len(t)} sampled workgroups, lifetime mean {life.mean():.0f} ticks (median {np.median(life):.0f}, p90 {np.percentile(life, 90):.0f})")
for pid, n in zip(order, names):
    dt = t[:, pid] - prev
    print(f"  {n:18s} {dt.mean():8.0f} ticks  {100 * dt.mean() / life.mean():5.1f} %   (median {np.median(dt):.0f})")
    prev = t[:, pid]
