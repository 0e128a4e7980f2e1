#!/usr/bin/env python3
"""Developer aid (GPU box): what a stage costs with six batches in flight -- the headline loop (32-frame batches of pyr3x8, 4 batch slots) with the stage sets
EXTRACT, EXTRACT | NMS and ALL: ms per batch each, so the differences are the stages' EXPOSED cost (to compare with their isolated kernel times)."""
import os, sys, threading, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch          # (first: the library must share torch's HIP runtime)
import str_er_amd as S
S.apply_runtime_hint()
F, P, W, H = int(os.environ.get("DEV_F", 32)), 6, 1920, 1080
S.set_batch_slots(4)
if os.environ.get("RAW"):        # the binding's copies of the result arrays left out (the C call has the records in host memory either way)
    def _raw(self, rh, profile=None):
        self.L.str_er_result_free(rh)
        return None
    S.ERFilter._collect = _raw
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
ctxs = []
for _ in range(P):
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=8, channel_mask=7))
    f.load_cascade(0, sp); f.load_cascade(1, wp)
    ctxs.append(f)
d = torch.from_numpy(S.synth.frames_bgr("text", 0, F, W, H)).cuda()


def run(stages, n):
    def work(p):
        for _ in range(n):
            try:
                ctxs[p].detect_bgr_device(d.data_ptr(), W, H, F, stages)
            except S.StrErError:
                pass            # (STR_ER_DEBUG_STOP_AFTER builds: the call ends after stage n)
    th = [threading.Thread(target=work, args=(p,)) for p in range(P)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * P)


for name, st in (("ALL", S.STAGE_ALL), ("EXTRACT | NMS", S.STAGE_EXTRACT | S.STAGE_NMS), ("EXTRACT", S.STAGE_EXTRACT), ("ALL", S.STAGE_ALL)):
    run(st, 5)
    t = run(st, 40)
    print(f"{name:14s} {1e3 * t:.3f} ms per {F}-frame batch = {F / t:.0f} frames/s", flush=True)
