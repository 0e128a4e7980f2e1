"""Parity of the configurations bench.py TIMES: the exact workloads of the bench line --
pyr3x8 (BASELINE configs[1], and configs[2] with the OCR scorer: 1920x1080 BGR, {Y,Cr,Cb} x 8 pyramid levels) and native6 (the reference's own six
planes) -- on the bench's own synthetic frames (S-text and S-noise, seed = 0x5EED0000 + frame index), every plane of every
frame compared with the oracle: node count, pool, classes, both cascade scores; config 3 adds the chain-code + SVM scorer.
Plus a bounded soak of the lock-free tree kernels (both sizes of the tile kernel) on random planes."""
import gzip
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import check_plane_against_oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 1920, 1080
TOL = 1e-4          # BASELINE.json north_star: classifier scores within 1e-4 (the cascade scores are compared exactly)


def _bench_frames(S):
    """2 S-text frames and 1 S-noise frame, seeded like bench.py's batch (global frame index 0, 1 / 0)."""
    return np.stack([S.synth.stext_bgr(S.synth.frame_seed(0), W, H), S.synth.stext_bgr(S.synth.frame_seed(1), W, H),
                     S.synth.snoise_bgr(S.synth.frame_seed(0), W, H)])


def _check_all_planes(oracle, oracle_cascades, res, plane_of):
    """Every plane of `res` against the oracle; the oracle runs on host threads (its C library releases the GIL)."""
    def one(p):
        check_plane_against_oracle(oracle, p, plane_of(p), oracle_cascades)
        return p.n_pool
    with ThreadPoolExecutor(8) as ex:
        return sum(ex.map(one, res.planes))


def test_bench_workload_pyr3x8(S, cascade_paths, oracle, oracle_cascades):
    """bench.py's default workload: all 24 planes of 2 S-text frames and 1 S-noise frame in ONE call, like a bench step."""
    levels, mask = 8, 0x07
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=3, n_pyr_levels=levels, channel_mask=mask))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frames = _bench_frames(S)
    res = f.text_detect(frames)
    assert len(res.planes) == 3 * 24
    assert [(p.frame, p.pyr, p.ch) for p in res.planes] == [(fr, l, c) for fr in range(3) for l in range(levels) for c in range(3)]
    pyr = {}
    for fr in range(3):
        six = oracle.compute_channels(frames[fr])
        for c in range(3):
            pyr[(fr, c)] = oracle.pyramid(six[c], levels)
    assert sum(p.width * p.height for p in res.planes[:24]) == 12395367       # the bytes the roofline counts
    n_pool = _check_all_planes(oracle, oracle_cascades, res, lambda p: pyr[(p.frame, p.ch)][p.pyr])
    assert n_pool == len(res.cands) > 1000
    assert sum(p.n_strong + p.n_weak for p in res.planes) > 100
    again = f.text_detect(frames)
    assert again.cands.tobytes() == res.cands.tobytes() and again.info.tobytes() == res.info.tobytes()       # deterministic
    f.close()


def test_bench_workload_ties(S, cascade_paths, oracle, oracle_cascades):
    """`bench.py --kind ties` / the `nms_ties_leg` of the default run: S-ties frames 2 and 5 (every third frame carries the double-L
    glyph).  Their Y planes have an NMS sibling tie with two different outcomes; the library must give the pool of the reference's
    flood order (the oracle's own child lists) -- and it must have needed the host walk for it."""
    levels, mask = 8, 0x07
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=3, n_pyr_levels=levels, channel_mask=mask))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frames = np.stack([S.synth.sties_bgr(S.synth.frame_seed(i), W, H, every=3) for i in (2, 4, 5)])
    before = f.tie_stats()["planes_walked"]
    res = f.text_detect(frames)
    pyr = {}
    for fr in range(3):
        six = oracle.compute_channels(frames[fr])
        for c in range(3):
            pyr[(fr, c)] = oracle.pyramid(six[c], levels)
    _check_all_planes(oracle, oracle_cascades, res, lambda p: pyr[(p.frame, p.ch)][p.pyr])
    st = f.tie_stats()
    assert st["planes_walked"] - before >= 2 and st["walk_ms_total"] > 0 and 1 <= st["host_threads"] <= 16
    # the tie is real: under the two key rules the oracle's pools of the glyph's plane differ
    t = oracle.tree_extract(pyr[(0, 0)][0])
    p1, _ = oracle.nms(t, H, W, sibling_mode=1)
    p2, a2 = oracle.nms(t, H, W, sibling_mode=2)
    assert a2 >= 1 and sorted(p1.tolist()) != sorted(p2.tolist())
    f.close()


def test_bench_workload_native6(S, cascade_paths, oracle, oracle_cascades):
    """`bench.py --workload native6`: the reference's six planes (src/ER.cpp:114-128), all 18 planes of the three frames."""
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=3))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frames = _bench_frames(S)
    res = f.text_detect(frames)
    assert len(res.planes) == 18
    six = [oracle.compute_channels(fr) for fr in frames]
    _check_all_planes(oracle, oracle_cascades, res, lambda p: six[p.frame][p.ch])
    f.close()


def test_bench_workload_config3_ocr(S, cascade_paths, oracle, oracle_cascades):
    """`bench.py --ocr` (BASELINE configs[2]) at 1920x1080: detect + classify + chain-code/SVM scorer on every strong/weak ER
    of a pyr3x8 S-text frame; detection compared exactly, OCR probabilities within 1e-4 of the oracle's libsvm restatement."""
    from oracle.oracle import OracleSVM
    levels, mask = 8, 0x07
    model = gzip.open(S.cascade_io.ocr_model_path()).read()
    import tempfile
    mp = os.path.join(tempfile.mkdtemp(), "ocr_synth.model")
    with open(mp, "wb") as fh:
        fh.write(model)
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, n_pyr_levels=levels, channel_mask=mask))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    f.load_svm_model_text(model, 1800)
    m = OracleSVM(oracle, mp)
    frame = S.synth.stext_bgr(S.synth.frame_seed(2), W, H)
    res = f.text_detect(frame, S.STAGE_ALL | S.STAGE_OCR)
    six = oracle.compute_channels(frame)
    pyr = {c: oracle.pyramid(six[c], levels) for c in range(3)}
    _check_all_planes(oracle, oracle_cascades, res, lambda p: pyr[p.ch][p.pyr])
    assert res.ocr_label is not None and len(res.ocr_label) == len(res.cands)
    n = 0
    for i, c in enumerate(res.cands):
        if c["cls"] == 0:
            assert res.ocr_label[i] == -1 and res.ocr_prob[i] == 0
            continue
        img = pyr[int(c["ch"])][int(c["pyr"])]
        q = oracle.chain_features(img[c["y"]:c["y"] + c["h"], c["x"]:c["x"] + c["w"]])
        l, p, _ = m.predict_probability(q / 255.0)
        assert abs(res.ocr_prob[i] - p.max()) < TOL
        top2 = np.sort(p)[-2:]
        if top2[1] - top2[0] > 10 * TOL:
            assert res.ocr_label[i] == l
        n += 1
    assert n > 100
    f.close()


@pytest.mark.parametrize("mode", ["sparse", "dense", "tile2"])
def test_soak_random_planes(S, oracle, monkeypatch, mode):
    """Bounded soak of the lock-free tree kernels (was tools/soak.py): random planes -- sizes up to 400x300, six value
    distributions, thresh steps 1-16, MIN_AREA 1/20/120 -- node for node against the oracle, for 25 s per tile-kernel size; "tile2": every
    plane through the second tile kernel (k_tile_tree2: level by level on bit masks), which hands the tiles it does not take -- on these
    planes a good part -- back to the first."""
    monkeypatch.setenv("STR_ER_TILE_KERNEL", "sparse" if mode == "tile2" else mode)
    monkeypatch.setenv("STR_ER_TILE2", "2" if mode == "tile2" else "0")
    # (STR_ER_SOAK_SECONDS / STR_ER_SOAK_SEED: a longer or different soak by hand)
    rng = np.random.default_rng({"sparse": 11, "dense": 12, "tile2": 13}[mode] + int(os.environ.get("STR_ER_SOAK_SEED", "0")))
    budget, t0, n = float(os.environ.get("STR_ER_SOAK_SECONDS", "25")), time.time(), 0
    filters, seen_t = {}, set()
    try:
        while time.time() - t0 < budget:
            step = int(rng.choice([1, 2, 4, 8, 8, 8, 16]))
            w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
            kind = int(rng.integers(0, 6))
            if kind == 5:
                # speckles of random density on a flat plane: tiles with anything from a handful to a thousand nodes (the small tile kernel changes
                # its way of working at 256 and at ~330 nodes per tile, both kernels at their fold capacity)
                img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
                m = rng.random((h, w)) < rng.uniform(0.01, 0.6)
                img[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
            elif kind == 0:
                img = rng.integers(0, 256, (h, w), dtype=np.uint8)
            elif kind == 1:
                img = (rng.integers(0, 2, (h, w)) * int(rng.integers(1, 255))).astype(np.uint8)
            elif kind == 2:
                base = np.add.outer(np.arange(h), np.arange(w)) * rng.uniform(0.05, 1.5)
                img = np.clip(base + rng.integers(-6, 7, (h, w)), 0, 255).astype(np.uint8)
            elif kind == 3:
                img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
                for _ in range(int(rng.integers(1, 30))):
                    x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
                    img[y0:y0 + int(rng.integers(1, 60)), x0:x0 + int(rng.integers(1, 60))] = int(rng.integers(0, 256))
            else:
                img = S.synth.gray(S.synth.stext_bgr(int(rng.integers(0, 1 << 30)), w, h)) if w >= 8 and h >= 8 else \
                    rng.integers(0, 256, (h, w), dtype=np.uint8)
            # the NMS parameters too (stability_t had only been run at 2 and 3, overlap_coef at 0.3 / 0.6 / 0.7): a pool of contexts per
            # thresh step, each with its own MIN_AREA, MAX_AREA, STABILITY_T in {0, 1, 2, 3, 5} and OVERLAP_COEF in [0.2, 0.9]
            # (inc/ER.h:113, src/ER.cpp:416-505; str_er_create accepts stability_t 0 .. 255)
            key = (step, int(rng.integers(0, 3)))
            if key not in filters:
                filters[key] = S.ERFilter(params=S.Params(thresh_step=step, min_area=int(rng.choice([1, 20, 120])), max_area=int(rng.choice([300, 5000, 900000])),
                                                          stability_t=int(rng.choice([0, 1, 2, 3, 5])), overlap_coef=float(np.round(rng.uniform(0.2, 0.9), 3)),
                                                          max_width=400, max_height=300, max_frames=1, kept_cap=130000, pool_cap=130000))
                seen_t.add(filters[key].params.stability_t)
            f = filters[key]
            p = f.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS, want_nodes=True).planes[0]
            check_plane_against_oracle(oracle, p, img, None, step=step, min_area=f.params.min_area, max_area=f.params.max_area,
                                       stability_t=f.params.stability_t, overlap_coef=f.params.overlap_coef)
            n += 1
    finally:
        for f in filters.values():
            f.close()
    print(f"soak[{mode}]: {n} planes in {time.time() - t0:.1f} s on {len(filters)} parameter sets (stability_t {sorted(seen_t)}), all equal to the oracle")
    assert n > 200 and len(seen_t) >= 3
