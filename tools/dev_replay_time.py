#!/usr/bin/env python3
"""Developer aid: how long does the flood replay (exact NMS tie-break) take on full-size planes?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import str_er_amd as S
from oracle.oracle import Oracle
o = Oracle()
for (w, h) in ((640, 480), (1920, 1080)):
    for kind in ("text", "noise"):
        img = S.synth.gray(S.synth.KINDS[kind](S.synth.frame_seed(3), w, h))
        for order in (2, 0):
            f = S.ERFilter(8, 30, 900000, 2, 0.3, max_width=w, max_height=h, max_frames=1, kept_cap=400000, pool_cap=100000, sibling_order=order)
            f.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS)
            t = time.perf_counter()
            r = f.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS)
            dt = time.perf_counter() - t
            print(f"{w}x{h} {kind} sibling_order={order}: {1e3 * dt:.1f} ms, ambiguous={r.planes[0].ambiguous}, pool={r.planes[0].n_pool}", flush=True)
            if order == 0:
                ref = o.detect_plane(img, None, None, min_area=30, overlap_coef=0.3)
                exp = sorted(int(ref["tree"].nodes[i]["key"]) for i in ref["pool"])
                assert exp == [int(c["key"]) for c in r.planes[0].cands], "mismatch"
            f.close()
