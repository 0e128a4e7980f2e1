import sys, importlib, numpy as np
sys.path.insert(0, '/root/repo')
S = importlib.import_module("scene-text-recognition_amd")
from oracle.oracle import Oracle
o = Oracle()
lut = o.quant_lut(8)
hi = o.highest_level(8)
for kind in ("text", "noise"):
    fn = S.synth.KINDS[kind]
    bgr = fn(S.synth.frame_seed(0), 1920, 1080)
    planes = o.compute_channels(bgr)
    for ci in (0, 1):
        lev = lut[planes[ci]].astype(np.uint8)
        lev[lev >= hi] = 255
        lev.tofile(f"{kind}_{ci}.lev")
        print(kind, ci, lev.shape, np.bincount(lev.ravel(), minlength=34)[:34])
    # a pyramid level
    pyr = o.pyramid(planes[0], 3)
    lev = lut[pyr[2]].astype(np.uint8); lev[lev >= hi] = 255
    lev.tofile(f"{kind}_p2.lev"); print(lev.shape)
