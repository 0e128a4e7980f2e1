import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import str_er_amd as S
from oracle.oracle import Oracle
o = Oracle()
tmp = tempfile.mkdtemp(); sp, wp = S.cascade_io.write_golden(tmp)
f = S.ERFilter(params=S.Params(max_width=320, max_height=240, max_frames=1, n_pyr_levels=4, channel_mask=0x0B))
f.load_cascade(0, sp); f.load_cascade(1, wp)
frame = S.synth.stext_bgr(9, 320, 240)
six = o.compute_channels(frame)
g6 = f.compute_channels(frame)
print('channels equal', (six == g6).all())
for ch in (0, 1, 3):
    pyr = o.pyramid(six[ch], 4)
    prev = six[ch]
    for l in range(1, 4):
        h, w = pyr[l].shape
        g = f.resize_plane(prev, w, h)
        print('ch', ch, 'level', l, (h, w), 'resize mismatches', int((g != pyr[l]).sum()))
        prev = pyr[l]
res = f.text_detect(frame, want_nodes=True)
for p in res.planes:
    img = o.pyramid(six[p.ch], 4)[p.pyr]
    t = o.tree_extract(img, 8, 120)
    r = t.nodes[t.root]
    gr = p.nodes[p.root]
    single = f.detect_planes(img, S.STAGE_EXTRACT, want_nodes=True).planes[0]
    sr = single.nodes[single.root]
    print('plane', p.ch, p.pyr, img.shape, 'root area oracle', r['area'], 'gpu-batch', gr['area'], 'gpu-single-on-oracle-plane', sr['area'], 'kept', len(t.nodes), p.n_kept, single.n_kept)
