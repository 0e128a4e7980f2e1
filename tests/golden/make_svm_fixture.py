#!/usr/bin/env python3
"""Mint the SVM fixtures with THE REFERENCE'S OWN libsvm (development container only).

classifier/OCR.model is missing from the reference checkout (.MISSING_LARGE_BLOBS), so a stand-in with
the same shape is trained with the reference's own svm-train (oracle/_ref/svm-train = src/svm-train.cpp +
src/svm.cpp compiled unmodified) and the reference's flags (`-b 1 -c 512 -g 0.0078125`,
src/utils.cpp:1544-1554): 65 classes, 1800 features in [0,1] (8 direction maps of 15x15, src/OCR.cpp:203-216).

Outputs:
  scene-text-recognition_amd/data/ocr_synth.model.gz   the libsvm text model (kept beside the cascades: bench.py --ocr uses it too)
                                      -- (svm_save_model format, src/svm.cpp:2641-2736)
  tests/golden/svm_vectors.npz      48 test vectors (8-bit numerators q, x = q/255.0 as in src/OCR.cpp:211) + what the reference's svm_predict_probability /
                                    svm_predict_values (oracle/_ref/libref_svm.so) return for them

`make_svm_fixture.py 120` mints the model AT THE REFERENCE'S TRAINING-SET SIZE: get_ocr_data (src/utils.cpp:1478-1541) writes one sample per class for
30 fonts x 4 styles = 120 samples per class, 7800 in all (the first stand-in has 5 per class: 319 support vectors, a kernel matrix of 320 columns and the
"at most 5 support vectors a class" build of k_svm_couple -- numbers measured on it flatter the scorer).  Same flags; samples noisier than the small set's
(sigma 0.3, 40 % of a prototype's features dropped, 60 stray features) so that, as with real glyph data, more than half of the training set ends up as
support vectors: 4299 of them, up to 90 per class.
  scene-text-recognition_amd/data/ocr_synth120.model.gz, tests/golden/svm_vectors120.npz
"""
import ctypes as C
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(ROOT, "oracle", "_ref")
K, D = 65, 1800
BIG = len(sys.argv) > 1 and sys.argv[1] == "120"
PER_CLASS = 120 if BIG else 5
NOISE, KEEP, EXTRA = (0.3, 0.6, 60) if BIG else (0.08, 0.85, 12)
MODEL_GZ = os.path.join(ROOT, "scene-text-recognition_amd", "data", "ocr_synth120.model.gz" if BIG else "ocr_synth.model.gz")
VECTORS = os.path.join(HERE, "svm_vectors120.npz" if BIG else "svm_vectors.npz")


def sample(rng, proto_idx, proto_val):
    keep = rng.random(len(proto_idx)) < KEEP
    idx = proto_idx[keep]
    val = np.clip(proto_val[keep] + rng.normal(0, NOISE, keep.sum()), 1 / 255.0, 1.0)
    extra = rng.choice(D, size=EXTRA, replace=False)
    x = np.zeros(D)
    x[idx] = np.round(val * 255) / 255.0          # features are v/255 (src/OCR.cpp:211)
    x[extra] = np.round(rng.uniform(0.05, 0.5, EXTRA) * 255) / 255.0
    return x


def main():
    assert os.path.exists(os.path.join(REF, "svm-train")), "run `make -C oracle` in the development container first"
    rng = np.random.default_rng(65)
    protos = []
    for c in range(K):
        idx = np.sort(rng.choice(D, size=rng.integers(90, 160), replace=False))
        protos.append((idx, rng.uniform(0.2, 1.0, len(idx))))
    tmp = tempfile.mkdtemp()
    data = os.path.join(tmp, "OCR.data")
    with open(data, "w") as f:
        for c in range(K):
            for _ in range(PER_CLASS):
                x = sample(rng, *protos[c])
                nz = np.nonzero(x)[0]
                f.write(str(c) + " " + " ".join(f"{i}:{x[i]:.8g}" for i in nz) + "\n")   # label + index:value (0-based like :209)
    model = os.path.join(tmp, "OCR.model")
    subprocess.run([os.path.join(REF, "svm-train"), "-q", "-b", "1", "-c", "512", "-g", "0.0078125", data, model], check=True)
    raw = open(model, "rb").read()
    with gzip.GzipFile(MODEL_GZ, "wb", mtime=0) as g:
        g.write(raw)

    class Node(C.Structure):
        _fields_ = [("index", C.c_int), ("value", C.c_double)]

    L = C.CDLL(os.path.join(REF, "libref_svm.so"))
    L.svm_load_model.restype = C.c_void_p
    L.svm_load_model.argtypes = [C.c_char_p]
    L.svm_predict_probability.restype = C.c_double
    L.svm_predict_probability.argtypes = [C.c_void_p, C.POINTER(Node), C.POINTER(C.c_double)]
    L.svm_predict_values.restype = C.c_double
    L.svm_predict_values.argtypes = [C.c_void_p, C.POINTER(Node), C.POINTER(C.c_double)]
    m = L.svm_load_model(model.encode())
    assert m
    n = 48
    X = np.zeros((n, D))
    for i in range(n):
        if i % 4 == 3:
            X[i, rng.choice(D, size=rng.integers(1, 300), replace=False)] = np.round(rng.uniform(0.01, 1, 1) * 255) / 255.0
        else:
            X[i] = sample(rng, *protos[rng.integers(0, K)])
    X[0] = 0                                             # the empty feature vector
    lab = np.zeros(n, np.int32)
    prob = np.zeros((n, K))
    dec = np.zeros((n, K * (K - 1) // 2))
    for i in range(n):
        nz = np.nonzero(X[i])[0]
        nodes = (Node * (len(nz) + 1))()
        for j, k in enumerate(nz):
            nodes[j].index, nodes[j].value = int(k), float(X[i, k])
        nodes[len(nz)].index = -1
        pv = (C.c_double * K)()
        dv = (C.c_double * (K * (K - 1) // 2))()
        lab[i] = int(L.svm_predict_probability(m, nodes, pv))
        L.svm_predict_values(m, nodes, dv)
        prob[i] = list(pv)
        dec[i] = list(dv)
    q = np.round(X * 255).astype(np.uint8)
    assert np.array_equal(q / 255.0, X), "features must be exact multiples of 1/255"
    np.savez_compressed(VECTORS, q=q, label=lab, prob=prob, dec=dec[:8])
    print("model", len(raw), "bytes ->", os.path.getsize(MODEL_GZ), "gz; vectors",
          os.path.getsize(VECTORS), "bytes; labels", np.bincount(lab, minlength=K).tolist()[:10], "...")


if __name__ == "__main__":
    sys.exit(main())
