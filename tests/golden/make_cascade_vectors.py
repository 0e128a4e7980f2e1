#!/usr/bin/env python3
"""Mint tests/golden/cascade_vectors.npz with THE REFERENCE ITSELF.

Development container only.  Loads oracle/_ref/libref_adaboost.so -- the reference's own
src/adaboost.cpp compiled unmodified (oracle/Makefile) -- and records what
CascadeBoost::predict (src/adaboost.cpp:507-542) returns for 256 synthetic Mean-LBP
histograms with the reference's trained strong/weak cascades.  These are reference outputs,
so they pin the oracle's and the HIP kernel's cascade arithmetic.

Histograms: 2x2 blocks of 144 LBP codes each (the shape make_LBP_hist produces,
src/ER.cpp:789-816); a mix of uniform-random codes, few-code (text-like) blocks and the
degenerate all-zero-code tile.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import RefCascade  # noqa: E402

REF = "/root/reference/classifier"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cascade_vectors.npz")


def main():
    assert RefCascade.available(), "needs /root/reference (development container)"
    strong = RefCascade(os.path.join(REF, "strong.classifier"))
    weak = RefCascade(os.path.join(REF, "weak.classifier"))
    rng = np.random.default_rng(20260927)
    hists = np.zeros((256, 1024), np.uint8)
    for i in range(256):
        for b in range(4):
            if i == 0:
                codes = np.zeros(144, np.int64)
            elif i % 3 == 0:
                codes = rng.integers(0, 256, 144)
            elif i % 3 == 1:
                pal = rng.choice(256, size=rng.integers(2, 9), replace=False)
                codes = rng.choice(pal, size=144)
            else:
                pal = np.array([0, 255, 15, 240, 31, 248, 7, 224, 1, 128])
                w = rng.dirichlet(np.ones(len(pal)) * 0.5)
                codes = rng.choice(pal, size=144, p=w)
            hists[i, b * 256:(b + 1) * 256] = np.bincount(codes, minlength=256)
    s = np.array([strong.predict(h.astype(np.float64)) for h in hists])
    w = np.array([weak.predict(h.astype(np.float64)) for h in hists])
    np.savez_compressed(OUT, hist=hists, strong=s, weak=w)
    fin = np.finfo(np.float64).max
    print("wrote", OUT, os.path.getsize(OUT), "bytes; strong accepts", int((s > -fin).sum()), "weak accepts", int((w > -fin).sum()))


if __name__ == "__main__":
    main()
