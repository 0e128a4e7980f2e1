#!/usr/bin/env python3
"""Developer timing: batched detect_bgr with device-resident frames, per-group profile."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import str_er_amd as S

def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    kind = sys.argv[2] if len(sys.argv) > 2 else 'text'
    levels = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    mask = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0x3F
    W, H = int(os.environ.get('DEV_W', 1920)), int(os.environ.get('DEV_H', 1080))
    tmp = tempfile.mkdtemp()
    sp, wp = S.cascade_io.write_golden(tmp)
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=levels, channel_mask=mask))
    f.set_profiling(True)
    f.load_cascade(0, sp); f.load_cascade(1, wp)
    print('workspace GB', f.workspace_bytes() / 1e9, flush=True)
    nsrc = min(F, 4)
    src = S.synth.frames_bgr(kind, 0, nsrc, W, H)
    frames = np.stack([src[i % nsrc] for i in range(F)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    for it in range(4):
        t0 = time.time()
        r = f.detect_bgr_device(d.data_ptr(), W, H, F)
        dt = time.time() - t0
        tot = sum(r.profile.values())
        print(f'iter {it}: wall {dt*1e3:.2f} ms  -> {F/dt:.1f} fps ; gpu groups {tot:.2f} ms; cands {len(r.cands)}; '
              + ' '.join(f'{k}={v:.3f}' for k, v in r.profile.items()), flush=True)
    print('tile2', f.tile2_stats(), flush=True)
    p = r.planes
    print('planes', len(p), 'kept', sum(x.n_kept for x in p), 'created', sum(x.n_created for x in p), 'pool', sum(x.n_pool for x in p),
          'strong', sum(x.n_strong for x in p), 'weak', sum(x.n_weak for x in p), 'amb', sum(x.ambiguous for x in p))

if __name__ == '__main__':
    main()
