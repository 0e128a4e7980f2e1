#!/usr/bin/env python3
"""Developer smoke check on a GPU box: HIP path vs oracle on a few planes."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import str_er_amd as S
from oracle.oracle import Oracle

o = Oracle()
tmp = tempfile.mkdtemp()
sp, wp = S.cascade_io.write_golden(tmp)
ostrong, oweak = o.cascade_load(sp), o.cascade_load(wp)


def canon_oracle(t):
    n = t.nodes
    return sorted((int(a['key']), int(a['level']), int(a['area']), int(a['x']), int(a['y']), int(a['w']), int(a['h']),
                   int(n[a['parent']]['key']) if a['parent'] >= 0 else int(a['key']),
                   int(n[a['parent']]['level']) if a['parent'] >= 0 else int(a['level'])) for a in n)


def canon_gpu(nodes):
    return sorted((int(a['key']), int(a['level']), int(a['area']), int(a['x']), int(a['y']), int(a['w']), int(a['h']),
                   int(nodes[a['parent']]['key']), int(nodes[a['parent']]['level'])) for a in nodes)


def check(f, img, tag, step=8, min_area=120):
    t0 = time.time()
    res = f.detect_planes(img, S.STAGE_ALL, want_nodes=True)
    t1 = time.time()
    p = res.planes[0]
    ref = o.detect_plane(img, ostrong, oweak, step=step, min_area=min_area)
    tr = ref['tree']
    ok_tree = canon_oracle(tr) == canon_gpu(p.nodes)
    rp = sorted((int(tr.nodes[i]['key']), int(tr.nodes[i]['level'])) for i in ref['pool'])
    gp = sorted((int(c['key']), int(c['level'])) for c in p.cands)
    ok_pool = rp == gp
    # classification
    order = np.argsort([int(tr.nodes[i]['key']) for i in ref['pool']], kind='stable')
    ok_cls = ok_pool and all(int(ref['cls'][j]) == int(c['cls']) and ref['s_strong'][j] == c['score_strong'] and
                             ref['s_weak'][j] == c['score_weak'] for j, c in zip(order, p.cands))
    print(f"{tag:28s} {img.shape} created {p.n_created}/{tr.n_created} kept {p.n_kept}/{len(tr.nodes)} pool {p.n_pool}/{len(ref['pool'])} "
          f"amb {p.ambiguous}/{ref['ambiguous']} strong {p.n_strong} weak {p.n_weak} tree={'OK' if ok_tree else 'BAD'} pool={'OK' if ok_pool else 'BAD'} "
          f"cls={'OK' if ok_cls else 'BAD'} gpu {1e3*(t1-t0):.1f} ms prof {res.profile}", flush=True)
    return ok_tree and ok_pool and ok_cls


def main():
    f = S.ERFilter(params=S.Params(max_width=1920, max_height=1080, max_frames=2))
    f.set_profiling(True)
    f.load_cascade(0, sp); f.load_cascade(1, wp)
    print('workspace MB', f.workspace_bytes() / 1e6, flush=True)
    rng = np.random.default_rng(1)
    ok = True
    ok &= check(f, np.full((10, 10), 100, np.uint8), 'const100', min_area=120)
    ok &= check(f, rng.integers(0, 256, (37, 53), dtype=np.uint8), 'noise small')
    ok &= check(f, rng.integers(0, 256, (64, 64), dtype=np.uint8), 'noise 64')
    ok &= check(f, rng.integers(0, 256, (100, 200), dtype=np.uint8), 'noise 100x200')
    ok &= check(f, S.synth.gray(S.synth.stext_bgr(3, 320, 240)), 'text 320x240')
    ok &= check(f, S.synth.gray(S.synth.stext_bgr(4, 640, 480)), 'text 640x480')
    ok &= check(f, S.synth.gray(S.synth.snoise_bgr(5, 640, 480)), 'noise 640x480')
    ok &= check(f, S.synth.gray(S.synth.stext_bgr(6, 1920, 1080)), 'text 1080p')
    ok &= check(f, S.synth.gray(S.synth.snoise_bgr(7, 1920, 1080)), 'noise 1080p')
    print('ALL OK' if ok else 'FAILURES')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
