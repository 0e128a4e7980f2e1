import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import str_er_amd as S
from oracle.oracle import Oracle
o = Oracle()
f = S.ERFilter(params=S.Params(max_width=640, max_height=480, max_frames=1))
rng = np.random.default_rng(3)
bgr = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
g = f.compute_channels(bgr); e = o.compute_channels(bgr)
for p in range(6):
    bad = (g[p] != e[p])
    print('plane', p, 'bad', bad.sum(), 'by x%4', [int(bad[:, k::4].sum()) for k in range(4)], 'by row<4', [int(bad[r].sum()) for r in range(4)])
d = (g[1].astype(int) - e[1].astype(int))
ys, xs = np.nonzero(d)
for y, x in list(zip(ys, xs))[:12]:
    B, G, R = [int(v) for v in bgr[y, x]]
    print((y, x), (B, G, R), 'gpu', g[1][y, x], 'exp', e[1][y, x], 'Ygpu', g[0][y, x], 'Cbgpu', g[2][y, x], 'Cbexp', e[2][y, x])
