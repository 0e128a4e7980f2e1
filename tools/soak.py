#!/usr/bin/env python3
"""Developer aid: soak test of the lock-free tree kernels -- many random planes (sizes, value distributions, thresh steps,
both sizes of the tile kernel), every one compared node for node with the oracle.  `python tools/soak.py [seconds] [seed]`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import str_er_amd as S
from oracle.oracle import Oracle
from conftest import check_plane_against_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = Oracle()
t0 = time.time()
n = 0
filters = {}
while time.time() - t0 < budget:
    mode = ("sparse", "dense")[n % 2]
    step = int(rng.choice([1, 2, 4, 8, 8, 8, 16]))
    w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1:
        img = (rng.integers(0, 2, (h, w)) * int(rng.integers(1, 255))).astype(np.uint8)
    elif kind == 2:
        base = np.add.outer(np.arange(h), np.arange(w)) * rng.uniform(0.05, 1.5)
        img = np.clip(base + rng.integers(-6, 7, (h, w)), 0, 255).astype(np.uint8)
    elif kind == 3:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        for _ in range(int(rng.integers(1, 30))):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            img[y0:y0 + int(rng.integers(1, 60)), x0:x0 + int(rng.integers(1, 60))] = int(rng.integers(0, 256))
    else:
        img = S.synth.gray(S.synth.stext_bgr(int(rng.integers(0, 1 << 30)), w, h)) if w >= 8 and h >= 8 else rng.integers(0, 256, (h, w), dtype=np.uint8)
    key = (mode, step)
    if key not in filters:
        os.environ["STR_ER_TILE_KERNEL"] = mode
        filters[key] = S.ERFilter(params=S.Params(thresh_step=step, min_area=int(rng.choice([1, 20, 120])), max_width=400, max_height=300, max_frames=1, kept_cap=130000, pool_cap=40000))
    f = filters[key]
    p = f.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS, want_nodes=True).planes[0]
    check_plane_against_oracle(oracle, p, img, None, step=step, min_area=f.params.min_area)
    n += 1
print(f"soak: {n} planes in {time.time() - t0:.1f} s, all equal to the oracle")
