#!/usr/bin/env python3
"""Mint scene-text-recognition_amd/data/cascades.npz (the model tables the package ships) from the reference's trained model files.

Run in the development container only (needs /root/reference):
    python tests/golden/make_cascades.py

The reference's classifier/strong.classifier and classifier/weak.classifier are
DATA (trained Real-AdaBoost cascades, format: SURVEY.md Appendix C /
src/adaboost.cpp:954-993).  This script parses them with Python's own float()
-- NOT with the oracle or the product parser -- and stores the numbers as
arrays, so the fixture is an independent statement of what the files hold.
`str_er_amd.cascade_io.write_classifier_text` turns the arrays back into the
reference's text format for the loader tests.
"""
import os
import sys

import numpy as np

REF = "/root/reference/classifier"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scene-text-recognition_amd", "data", "cascades.npz")


def parse(path):
    with open(path) as f:
        lines = f.read().split("\n")
    assert lines[0].split() == ["boost_type", "REAL"], lines[0]
    assert lines[1].split() == ["base_type", "DECISION_STUMP"], lines[1]
    it = lines[2].split()
    th = lines[3].split()
    assert it[0] == "num_of_iter" and th[0] == "threshold"
    stage_n = np.array([int(v) for v in it[1:]], np.int32)
    stage_t = np.array([int(float(v)) for v in th[1:]], np.int32)
    rows = [l.split() for l in lines[4:] if l.strip()]
    arr = np.array([[float(v) for v in r] for r in rows], np.float64)
    assert arr.shape == (int(stage_n.sum()), 5), arr.shape
    return stage_n, stage_t, arr


def main():
    out = {}
    for name in ("strong", "weak"):
        stage_n, stage_t, arr = parse(os.path.join(REF, f"{name}.classifier"))
        out[f"{name}_stage_n"] = stage_n
        out[f"{name}_stage_thresh"] = stage_t
        out[f"{name}_weight"] = arr[:, 0]
        out[f"{name}_dim"] = arr[:, 1].astype(np.int32)
        out[f"{name}_thresh"] = arr[:, 2]
        out[f"{name}_cp"] = arr[:, 3]
        out[f"{name}_cn"] = arr[:, 4]
        print(name, "stages", stage_n.tolist(), "thresh", stage_t.tolist(), "stumps", len(arr))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
