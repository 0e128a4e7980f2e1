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
t = np.array(out[:], dtype=np.float64).reshape(512, 16)[384:]
t = t[(t[:, 0] > 0) & (t[:, 4] > 0)]
print("classify ms", r.profile["classify"], "sampled workgroups", len(t))
life = t[:, 4] - t[:, 0]
print(f"lifetime mean {life.mean():.0f} ticks")
for i, n in enumerate(["phase 1 (LBP histograms)", "(A,B) tables into LDS", "stage sums", "cascade rule + records"]):
    dt = t[:, i + 1] - t[:, i]
    print(f"  {n:28s} {dt.mean():9.0f} ticks {100 * dt.mean() / life.mean():5.1f} %")
for a, b, n in ((5, 6, "first ER: zero the scratch"), (6, 7, "first ER: ARAN resize"), (7, 8, "first ER: LBP + histogram"), (8, 9, "first ER: pack the row")):
    dt = t[:, b] - t[:, a]
    print(f"  {n:28s} {dt.mean():9.0f} ticks")
try:
    c = r.cands
    a = (c["w"].astype(np.int64) * c["h"])
    print("candidates", len(c), "box area percentiles 10/50/90/99:", np.percentile(a, [10, 50, 90, 99]).astype(int), " share with w*h <= 4096:", float((a <= 4096).mean()),
          " w<=26&h<=26:", float(((c["w"] <= 26) & (c["h"] <= 26)).mean()))
except Exception as e:
    print("no cands:", type(r), [k for k in dir(r) if not k.startswith("_")][:20], e)
