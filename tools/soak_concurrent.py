#!/usr/bin/env python3
"""Developer aid (GPU box): the lock-free tree passes under the load they see in bench.py -- P contexts in flight on one GPU, every result compared
byte for byte with the result of the same batch computed alone beforehand (which tests/test_bench_workload.py ties to the oracle).
`python tools/soak_concurrent.py [seconds] [frames per batch] [contexts]`"""
import os, sys, threading, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
P = int(sys.argv[3]) if len(sys.argv) > 3 else 6
W, H = 1920, 1080
tmp = tempfile.mkdtemp(); sp, wp = S.cascade_io.write_golden(tmp)
S.set_batch_slots(int(os.environ.get("SLOTS", "0")))


def make():
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=8, channel_mask=7))
    f.load_cascade(0, sp); f.load_cascade(1, wp)
    return f


NB = 4
batches = []
for b in range(NB):
    kinds = ["text"] * F
    if b == 1:
        kinds[F // 2] = "noise"
    fr = np.stack([S.synth.frames_bgr(k, 1000 * b + i, 1, W, H)[0] for i, k in enumerate(kinds)])
    batches.append(torch.from_numpy(fr).cuda())
ref = make()
expected = []
for d in batches:
    r = ref.detect_bgr_device(d.data_ptr(), W, H, F)
    r2 = ref.detect_bgr_device(d.data_ptr(), W, H, F)
    assert r.cands.tobytes() == r2.cands.tobytes() and r.info.tobytes() == r2.info.tobytes()
    expected.append((r.cands.tobytes(), r.info.tobytes(), len(r.cands)))
print("reference results:", [e[2] for e in expected], "candidates per batch", flush=True)
ctxs = [make() for _ in range(P)]
bad, done = [], [0] * P
t_end = time.time() + budget


def work(p):
    k = p
    while time.time() < t_end and not bad:
        b = k % NB
        r = ctxs[p].detect_bgr_device(batches[b].data_ptr(), W, H, F)
        if r.cands.tobytes() != expected[b][0] or r.info.tobytes() != expected[b][1]:
            bad.append((p, k, b, len(r.cands)))
        k += 1; done[p] += 1


th = [threading.Thread(target=work, args=(p,)) for p in range(P)]
for t in th: t.start()
for t in th: t.join()
print(f"{sum(done)} batches of {F} frames over {P} contexts in {budget:.0f} s: {'MISMATCH ' + str(bad) if bad else 'all equal to the quiet run'}")
sys.exit(1 if bad else 0)
