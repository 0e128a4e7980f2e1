#!/usr/bin/env python3
"""Developer aid: one frame per call, one call in flight (bench.py's latency_1frame): wall time per call and the library's own per-stage GPU times."""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S
W, H = 1920, 1080
tmp = tempfile.mkdtemp(); sp, wp = S.cascade_io.write_golden(tmp)
f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, n_pyr_levels=8, channel_mask=7))
f.load_cascade(0, sp); f.load_cascade(1, wp)
frames = S.synth.frames_bgr("text", 0, 8, W, H)
d = torch.from_numpy(frames).cuda()
fb = frames[0].nbytes
n = int(os.environ.get("DEV_LAT_N", "40"))
for prof_on in (False, True):          # (the per-group events cost stream time: the call is timed without them first)
    f.set_profiling(prof_on)
    for i in range(5):
        f.detect_bgr_device(d.data_ptr() + (i % 8) * fb, W, H, 1)
    ms, prof = [], {}
    for i in range(n):
        t0 = time.perf_counter()
        r = f.detect_bgr_device(d.data_ptr() + (i % 8) * fb, W, H, 1)
        ms.append(1e3 * (time.perf_counter() - t0))
        for k, v in r.profile.items():
            prof[k] = prof.get(k, 0.0) + v / n
    print(f"1 frame per call ({'with' if prof_on else 'without'} per-group events): median {np.median(ms):.3f} ms, p10 {np.percentile(ms, 10):.3f}, p90 {np.percentile(ms, 90):.3f}")
print("GPU ms per stage (events):", {k: round(v, 4) for k, v in prof.items()}, "sum", round(sum(prof.values()), 4))
