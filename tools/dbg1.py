import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import str_er_amd as S
from oracle.oracle import Oracle, RefCascade
import tempfile
o = Oracle()
f = S.ERFilter(params=S.Params(max_width=640, max_height=480, max_frames=1))
rng = np.random.default_rng(3)
for shape in ((1, 1, 3), (7, 13, 3), (48, 64, 3), (33, 101, 3)):
    bgr = rng.integers(0, 256, shape, dtype=np.uint8)
    g = f.compute_channels(bgr); e = o.compute_channels(bgr)
    bad = np.argwhere(g != e)
    print(shape, 'mismatches', len(bad), bad[:5].tolist())
    for b in bad[:3]:
        print('  px', bgr[b[1], b[2]].tolist(), 'gpu', g[b[0], b[1], b[2]], 'oracle', e[b[0], b[1], b[2]])
tmp = tempfile.mkdtemp(); sp, wp = S.cascade_io.write_golden(tmp)
f.load_cascade(0, sp); f.load_cascade(1, wp)
rs = RefCascade(sp)
fv = np.stack([np.concatenate([np.bincount(rng.integers(0, 256, 144) // rng.integers(1, 40), minlength=256) for _ in range(4)]) for _ in range(300)]).astype(np.float64)
gs = f.predict(0, fv); es = np.array([rs.predict(v) for v in fv])
bad = np.argwhere(gs != es).ravel()
print('cascade mismatches', len(bad), [(gs[i], es[i]) for i in bad[:5]], 'accepts', (es > -1e300).sum())
