#!/usr/bin/env python3
"""Fuzz loop over the entry points that take untrusted bytes (run under tools/san_run.sh asan on a GPU box):
   str_er_load_cascade_mem (the reference's text format), str_er_load_svm_model_mem (libsvm text), str_er_strip_merge (strip blobs from
   another process).  Every mutated input must come back as an error code or as a valid load -- never a crash, a sanitizer report or an
   out-of-range device access -- and the context must stay usable.   python tools/san_fuzz.py [seconds] [seed]"""
import gzip, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import str_er_amd as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, H = 320, 200
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
texts = [open(sp).read().encode(), open(wp).read().encode()]
svm = gzip.open(S.cascade_io.ocr_model_path()).read()
frame = S.synth.stext_bgr(S.synth.frame_seed(3), W, H)


def mutate(b: bytes) -> bytes:
    a = bytearray(b)
    kind = int(rng.integers(0, 6))
    if kind == 0 and len(a) > 8:                    # truncate
        del a[int(rng.integers(1, len(a))):]
    elif kind == 1:                                  # flip a few bytes
        for _ in range(int(rng.integers(1, 8))):
            a[int(rng.integers(0, len(a)))] = int(rng.integers(0, 256))
    elif kind == 2:                                  # overwrite a 32-bit word with an extreme value
        if len(a) >= 8:
            at = int(rng.integers(0, len(a) - 4)) & ~3
            a[at:at + 4] = int(rng.choice([0, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0x00FFFFF0, 1 << 24])).to_bytes(4, "little")
    elif kind == 3:                                  # duplicate a slice
        i, j = sorted(int(v) for v in rng.integers(0, len(a), 2))
        a[i:i] = a[i:j][:4096]
    elif kind == 4:                                  # replace a number token of a text by something odd
        toks = a.split(b" ")
        if len(toks) > 4:
            toks[int(rng.integers(0, len(toks)))] = rng.choice([b"nan", b"-1", b"1e999", b"99999999999", b"", b"x", b"0x10"])
            a = bytearray(b" ".join(toks))
    else:                                            # drop a line
        lines = a.split(b"\n")
        if len(lines) > 2:
            del lines[int(rng.integers(0, len(lines)))]
            a = bytearray(b"\n".join(lines))
    return bytes(a)


def make():
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, kept_cap=40000, pool_cap=10000))
    f.load_cascade(0, sp); f.load_cascade(1, wp)
    return f


owner = make()
workers = [make() for _ in range(3)]
blobs = [workers[s].strip_extract(frame, s, 3) for s in range(3)]
good = owner.strip_merge(frame, blobs).cands.tobytes()
t0, n, n_err = time.time(), 0, 0
while time.time() - t0 < budget:
    which = n % 3
    if os.environ.get("SAN_FUZZ_TRACE"):
        print(n, which, flush=True)
    try:
        if which == 0:
            k = int(rng.integers(0, 2))
            owner.load_cascade_text(k, mutate(texts[k]).decode("latin-1"))
        elif which == 1:
            owner.load_svm_model_text(mutate(svm[: int(rng.integers(2000, 200000))]), 1800)
        else:
            i = int(rng.integers(0, 3))
            bl = list(blobs)
            bl[i] = mutate(blobs[i])
            owner.strip_merge(frame, bl)
    except S.StrErError:
        n_err += 1
    n += 1
    if n % 50 == 0:         # the context still works, with the right answer
        owner.load_cascade(0, sp); owner.load_cascade(1, wp)
        assert owner.strip_merge(frame, blobs).cands.tobytes() == good
owner.load_cascade(0, sp); owner.load_cascade(1, wp)
assert owner.strip_merge(frame, blobs).cands.tobytes() == good
print(f"san_fuzz: {n} mutated inputs in {time.time() - t0:.1f} s, {n_err} rejected with an error code, {n - n_err} accepted; context intact")
