"""Text format of the reference's cascade files.

`write_classifier_text` produces what CascadeBoost::write_classifier writes
(src/adaboost.cpp:954-993): header lines, then one `weight dim thresh cp cn ` row per
stump (each row ends with a space).  Numbers are printed with repr()-precision so that
strtod() gives back exactly the doubles stored in data/cascades.npz (the reference's trained
strong / weak cascades as arrays; minted by tests/golden/make_cascades.py).
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cascades.npz")
OCR_MODEL_GZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ocr_synth.model.gz")


OCR_MODEL120_GZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ocr_synth120.model.gz")


def ocr_model_path(per_class: int = 5) -> str:
    """The gzip-ed libsvm text model that stands in for the reference's missing classifier/OCR.model: 65 classes, 1800 features,
    trained on synthetic vectors by the reference's own svm-train with the reference's flags (tests/golden/make_svm_fixture.py).
    per_class = 120: the reference's training-set size (30 fonts x 4 styles, src/utils.cpp:1478-1541), 4299 support vectors; per_class = 5: the
    small model of rounds 1-5 (319 support vectors) that the parity tests were written on."""
    if per_class not in (5, 120):
        raise ValueError("the stand-in OCR models have 5 or 120 samples per class")
    return OCR_MODEL120_GZ if per_class == 120 else OCR_MODEL_GZ


def _num(v: float) -> str:
    v = float(v)
    return str(int(v)) if v == int(v) and abs(v) < 1e15 else repr(v)


def classifier_text(stage_n, stage_thresh, weight, dim, thresh, cp, cn) -> str:
    lines = ["boost_type REAL", "base_type DECISION_STUMP",
             "num_of_iter " + " ".join(str(int(v)) for v in stage_n),
             "threshold " + " ".join(str(int(v)) for v in stage_thresh)]
    for i in range(len(dim)):
        lines.append(f"{_num(weight[i])} {int(dim[i])} {_num(thresh[i])} {_num(cp[i])} {_num(cn[i])} ")
    return "\n".join(lines) + "\n"


def golden_text(which: str, npz_path: str = GOLDEN) -> str:
    """Text of the reference's trained `strong` / `weak` cascade, rebuilt from the fixture."""
    z = np.load(npz_path)
    return classifier_text(z[f"{which}_stage_n"], z[f"{which}_stage_thresh"], z[f"{which}_weight"], z[f"{which}_dim"],
                           z[f"{which}_thresh"], z[f"{which}_cp"], z[f"{which}_cn"])


def write_golden(dirpath: str, npz_path: str = GOLDEN):
    """Write strong.classifier / weak.classifier into dirpath; returns the two paths."""
    os.makedirs(dirpath, exist_ok=True)
    out = []
    for which in ("strong", "weak"):
        p = os.path.join(dirpath, f"{which}.classifier")
        with open(p, "w") as f:
            f.write(golden_text(which, npz_path))
        out.append(p)
    return tuple(out)
