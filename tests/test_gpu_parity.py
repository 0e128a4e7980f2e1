"""GPU parity tests: the HIP path, called through the C ABI, against the oracle on the same
inputs.  Integer/index outputs (node tables, pools, classes) are compared bit for bit; the
cascade scores are compared EXACTLY as well (the kernel adds stump outputs in file order),
which is stricter than the 1e-4 tolerance BASELINE.json allows."""
import os

import numpy as np
import pytest

from conftest import check_plane_against_oracle, gpu_tree_canon, oracle_tree_canon

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DBL_MAX = float(np.finfo(np.float64).max)


def _planes(rng, shapes):
    for k, (h, w) in enumerate(shapes):
        mode = k % 5
        if mode == 0:
            yield "noise", rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif mode == 1:
            yield "4lev", (rng.integers(0, 4, (h, w)) * 60).astype(np.uint8)
        elif mode == 2:
            yield "sentinel", rng.integers(200, 256, (h, w)).astype(np.uint8)
        elif mode == 3:
            yield "ramp", (np.add.outer(np.arange(h) * 3, np.arange(w) * 5) % 256).astype(np.uint8)
        else:
            yield "blocks", np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8), dtype=np.uint8), np.ones((8, 8), np.uint8))[:h, :w]


SHAPES = [(1, 1), (1, 17), (23, 1), (2, 2), (31, 63), (32, 64), (33, 65), (64, 128), (65, 129), (70, 130), (97, 211),
          (128, 64), (5, 300), (300, 5), (200, 333)]


@pytest.mark.parametrize("min_area", [0, 120])
def test_planes_small_all_nodes(erf, oracle, oracle_cascades, min_area):
    """Ragged sizes around the 64x32 tile; min_area=0 keeps (and so checks) every tree node."""
    rng = np.random.default_rng(100 + min_area)
    erf.set_min_area(min_area)
    try:
        for name, img in _planes(rng, SHAPES):
            res = erf.detect_planes(img, want_nodes=True)
            check_plane_against_oracle(oracle, res.planes[0], img, oracle_cascades, min_area=min_area)
    finally:
        erf.set_min_area(120)


def test_known_answers(erf):
    """SURVEY.md Appendix B known-answer runs of the unmodified reference."""
    p = erf.er_tree_extract(np.full((10, 10), 100, np.uint8))
    n = p.nodes
    assert len(n) == 1 and (n[0]["level"], n[0]["area"], n[0]["w"], n[0]["h"]) == (12, 101, 10, 10)
    p = erf.er_tree_extract(np.full((10, 10), 252, np.uint8))
    n = p.nodes
    assert len(n) == 1 and (n[0]["level"], n[0]["area"], n[0]["x"], n[0]["y"], n[0]["w"], n[0]["h"]) == (32, 2, 0, 0, 1, 1)
    m = np.full((6, 6), 40, np.uint8)
    m[0, 0] = 255
    assert erf.er_tree_extract(m).nodes[0]["area"] == 36
    m[0, 1] = 255
    assert erf.er_tree_extract(m).nodes[0]["area"] == 35
    m[1, 0] = 255
    n = erf.er_tree_extract(m).nodes
    assert (n[0]["level"], n[0]["area"], n[0]["w"], n[0]["h"]) == (32, 2, 1, 1)


def test_sentinel_walls(erf, oracle, oracle_cascades):
    """Closed curves of >=252 pixels hide what is behind them, across tiles."""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 250, (150, 260), dtype=np.uint8)
    img[20:130, 40] = 255; img[20:130, 200] = 253; img[20, 40:201] = 252; img[129, 40:201] = 254   # sealed box
    img[60:90, 100:140] = rng.integers(0, 250, (30, 40))
    erf.set_min_area(0)
    try:
        res = erf.detect_planes(img, want_nodes=True)
        ref = check_plane_against_oracle(oracle, res.planes[0], img, oracle_cascades, min_area=0)
        assert int(ref["tree"].nodes[ref["tree"].root]["npix"]) < img.size - 110 * 160 + 1000
        img[0, 0] = 255                                           # start pixel itself is a wall
        res = erf.detect_planes(img, want_nodes=True)
        check_plane_against_oracle(oracle, res.planes[0], img, oracle_cascades, min_area=0)
        img[0, 1] = 255
        img[1, 0] = 255                                           # ... and both candidates: single node
        res = erf.detect_planes(img, want_nodes=True)
        assert res.planes[0].n_kept == 1
        check_plane_against_oracle(oracle, res.planes[0], img, oracle_cascades, min_area=0)
    finally:
        erf.set_min_area(120)


@pytest.mark.parametrize("step,min_area", [(1, 30), (2, 50), (9, 20), (16, 200)])
def test_other_thresh_steps(S, cascade_paths, oracle, oracle_cascades, step, min_area):
    """THRESH_STEP and MIN_AREA are run-time parameters (src/utils.cpp:680, 716, 940, 1090)."""
    f = S.ERFilter(step, min_area, 900000, 2, 0.7, max_width=256, max_height=256, max_frames=1)
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    rng = np.random.default_rng(step)
    for img in (rng.integers(0, 256, (90, 150), dtype=np.uint8), S.synth.gray(S.synth.stext_bgr(step, 256, 200))):
        res = f.detect_planes(img, want_nodes=True)
        check_plane_against_oracle(oracle, res.planes[0], img, oracle_cascades, step=step, min_area=min_area)
    f.close()


def test_nms_parameters(S, cascade_paths, oracle, oracle_cascades):
    f = S.ERFilter(8, 40, 5000, 3, 0.6, max_width=320, max_height=240, max_frames=1)
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    img = S.synth.gray(S.synth.stext_bgr(21, 320, 240))
    res = f.detect_planes(img, want_nodes=True)
    check_plane_against_oracle(oracle, res.planes[0], img, oracle_cascades, min_area=40, max_area=5000, stability_t=3,
                               overlap_coef=0.6)
    f.close()


def test_batch_of_planes_is_independent(erf, oracle, oracle_cascades, S):
    imgs = np.stack([S.synth.gray(S.synth.stext_bgr(30 + i, 320, 240)) for i in range(5)])
    res = erf.detect_planes(imgs, want_nodes=True)
    assert len(res.planes) == 5
    for p, img in zip(res.planes, imgs):
        check_plane_against_oracle(oracle, p, img, oracle_cascades)
    single = erf.detect_planes(imgs[3], want_nodes=True).planes[0]
    fields = ["level", "cls", "x", "y", "w", "h", "area", "key", "score_strong", "score_weak"]
    assert single.cands[fields].tolist() == res.planes[3].cands[fields].tolist()


def test_bgr_frame_all_six_planes(erf, oracle, oracle_cascades, S):
    """ERFilter::text_detect up to classify on a 640x480 frame (src/ER.cpp:39-60)."""
    frame = S.synth.stext_bgr(S.synth.frame_seed(1), 640, 480)
    res = erf.text_detect(frame, want_nodes=True)
    planes = oracle.compute_channels(frame)
    assert [p.ch for p in res.planes] == [0, 1, 2, 3, 4, 5]
    for p in res.planes:
        check_plane_against_oracle(oracle, p, planes[p.ch], oracle_cascades)
    assert sum(p.n_strong + p.n_weak for p in res.planes) > 0
    assert res.times.shape == (7,) and res.times[0] > 0 and res.times[6] >= res.times[0]


def test_last_tree_stats(erf, S):
    """str_er_last_tree_stats (measurement aid for bench.py's tree_passes_roofline): tiles and border pairs follow from the plane sizes, the
    record count is what the tile kernel exported."""
    frame = S.synth.stext_bgr(S.synth.frame_seed(1), 640, 480)
    erf.text_detect(frame)
    st = erf.last_tree_stats()
    tx, ty = (640 + 63) // 64, (480 + 31) // 32
    assert st["tiles"] == 6 * tx * ty
    assert st["seam_pairs"] == 6 * ((ty - 1) * 640 + (tx - 1) * 480)
    assert 6 * tx * ty <= st["records"] < 6 * 640 * 480


def test_compute_channels(erf, oracle, S):
    rng = np.random.default_rng(3)
    for shape in ((1, 1, 3), (7, 13, 3), (48, 64, 3), (33, 101, 3)):
        bgr = rng.integers(0, 256, shape, dtype=np.uint8)
        assert (erf.compute_channels(bgr) == oracle.compute_channels(bgr)).all()
    corners = np.array([[[b, g, r] for b in (0, 255) for g in (0, 255) for r in (0, 255)]], np.uint8)
    assert (erf.compute_channels(corners) == oracle.compute_channels(corners)).all()


def test_resize_plane(erf, oracle):
    rng = np.random.default_rng(4)
    for (sh, sw), (dh, dw) in (((108, 192), (76, 136)), ((52, 52), (26, 26)), ((52, 48), (26, 24)), ((30, 30), (30, 30)),
                               ((9, 200), (26, 5)), ((200, 9), (8, 26)), ((3, 3), (26, 26)), ((1, 1), (7, 5))):
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        assert (erf.resize_plane(src, dw, dh) == oracle.resize(src, dw, dh)).all(), (sh, sw, dh, dw)


def test_pyramid_planes(S, cascade_paths, oracle, oracle_cascades):
    """Build-defined pyramid: level k = resize_linear(level k-1) of Y/Cr/Cb; an inverted channel at
    level k is 255 - (level k of its source channel).  Every level is checked as a plane."""
    f = S.ERFilter(params=S.Params(max_width=320, max_height=240, max_frames=1, n_pyr_levels=4, channel_mask=0x0B))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frame = S.synth.stext_bgr(9, 320, 240)
    res = f.text_detect(frame, want_nodes=True)
    six = oracle.compute_channels(frame)
    assert [(p.pyr, p.ch) for p in res.planes] == [(l, c) for l in range(4) for c in (0, 1, 3)]
    pyr = {c: oracle.pyramid(six[c], 4) for c in (0, 1)}
    pyr[3] = [255 - p for p in pyr[0]]
    for p in res.planes:
        img = pyr[p.ch][p.pyr]
        assert (p.height, p.width) == img.shape
        check_plane_against_oracle(oracle, p, img, oracle_cascades)
    f.close()


def test_inverted_channel_equals_materialised_inverse(erf, S):
    frame = S.synth.stext_bgr(12, 320, 240)
    res = erf.text_detect(frame, want_nodes=True)
    six = erf.compute_channels(frame)
    for ch in (3, 4, 5):
        direct = erf.detect_planes(six[ch], want_nodes=True).planes[0]
        assert gpu_tree_canon(direct.nodes) == gpu_tree_canon(res.planes[ch].nodes)
        fields = ["level", "cls", "x", "y", "w", "h", "area", "key", "score_strong", "score_weak"]
        assert direct.cands[fields].tolist() == res.planes[ch].cands[fields].tolist()


# ---- single stages --------------------------------------------------------------------------------
def _boxes(rng, h, w, n):
    out = [(0, 0, 52, 52), (3, 2, 48, 52), (0, 0, w, h), (5, 5, 26, 26), (1, 1, 3, 3), (0, 0, 1, 1), (10, 0, 200, 21), (0, 10, 9, 88)]
    for _ in range(n):
        bw, bh = int(rng.integers(1, min(w, 230))), int(rng.integers(1, min(h, 230)))
        out.append((int(rng.integers(0, w - bw + 1)), int(rng.integers(0, h - bh + 1)), bw, bh))
    return np.array([b for b in out if b[0] + b[2] <= w and b[1] + b[3] <= h], np.int32)


def test_lbp_hist_and_tiles(erf, oracle):
    """make_LBP_hist / calc_LBP / ARAN (src/ER.cpp:789-845, src/OCR.cpp:394-430) on assorted ROIs,
    including the exact-2x (52x52, 48x52) resize path and extreme aspect ratios."""
    rng = np.random.default_rng(5)
    plane = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    boxes = _boxes(rng, 240, 320, 120)
    hist, tiles = erf.make_LBP_hist(plane, boxes, return_tiles=True)
    for b, h, t in zip(boxes, hist, tiles):
        roi = plane[b[1]:b[1] + b[3], b[0]:b[0] + b[2]]
        assert (t == oracle.aran26(roi)).all(), b
        assert (h == oracle.lbp_hist(roi)).all(), b
    whole = erf.make_LBP_hist(plane)
    assert (whole[0] == oracle.lbp_hist(plane)).all()
    # Mat ERFilter::calc_LBP (inc/ER.h:134): the 24x24 code maps themselves, same ROIs
    codes = erf.calc_LBP(plane, boxes)
    for b, c in zip(boxes, codes):
        roi = plane[b[1]:b[1] + b[3], b[0]:b[0] + b[2]]
        assert (c == oracle.lbp24(oracle.aran26(roi))).all(), b
    assert (erf.calc_LBP(plane)[0] == oracle.lbp24(oracle.aran26(plane))).all()


def test_classify_boxes(erf, oracle, oracle_cascades, S):
    img = S.synth.gray(S.synth.stext_bgr(14, 640, 480))
    rng = np.random.default_rng(6)
    boxes = _boxes(rng, 480, 640, 150)
    cls, ss, sw = erf.classify(img, boxes)
    ecls, ess, esw = oracle.classify(img, boxes, *oracle_cascades)
    assert (cls == ecls).all() and (ss == ess).all() and (sw == esw).all()


def test_cascade_predict_matches_reference_vectors(erf):
    """tests/golden/cascade_vectors.npz = outputs of the reference's own CascadeBoost::predict."""
    z = np.load(os.path.join(GOLDEN, "cascade_vectors.npz"))
    fv = z["hist"].astype(np.float64)
    assert (erf.predict(0, fv) == z["strong"]).all()
    assert (erf.predict(1, fv) == z["weak"]).all()


def test_cascade_predict_matches_reference_library(erf, cascade_paths):
    from oracle.oracle import RefCascade
    if not RefCascade.available():
        pytest.skip("oracle/_ref/libref_adaboost.so not present")
    rs, rw = RefCascade(cascade_paths[0]), RefCascade(cascade_paths[1])
    rng = np.random.default_rng(8)
    fv = np.stack([np.concatenate([np.bincount(rng.integers(0, 256, 144) // rng.integers(1, 40), minlength=256)
                                   for _ in range(4)]) for _ in range(300)]).astype(np.float64)
    gs, gw = erf.predict(0, fv), erf.predict(1, fv)
    assert (gs == np.array([rs.predict(v) for v in fv])).all()
    assert (gw == np.array([rw.predict(v) for v in fv])).all()


def _to_node_table(S, tree):
    n = tree.nodes
    out = np.zeros(len(n), S.NODE_DTYPE)
    out["key"], out["area"], out["level"] = n["key"], n["area"], n["level"]
    out["x"], out["y"], out["w"], out["h"] = n["x"], n["y"], n["w"], n["h"]
    out["parent"] = np.where(n["parent"] < 0, np.arange(len(n)), n["parent"])
    return out


def test_nms_tree(erf, oracle, S):
    """non_maximum_supression on uploaded trees, incl. a hand-made ambiguous one."""
    for seed, kind in ((1, "text"), (2, "noise"), (3, "text")):
        img = S.synth.gray(S.synth.KINDS[kind](seed, 400, 300))
        t = oracle.tree_extract(img, 8, 120)
        ref, ramb = oracle.nms(t, 300, 400)
        # with the plane: ties are decided by the replayed flood order = the oracle's child lists (sibling_mode 0)
        pool, amb = erf.non_maximum_supression(_to_node_table(S, t), 300, 400, plane=img)
        assert (amb == 0) == (ramb == 0)
        assert sorted(pool.tolist()) == sorted(ref.tolist())
        assert [int(t.nodes[i]["key"]) for i in pool] == sorted(int(t.nodes[i]["key"]) for i in pool)
    # two children whose boxes both cover > 0.7 of the parent: the answer depends on sibling order
    from oracle.oracle import NODE_DTYPE, Tree
    n = np.zeros(5, NODE_DTYPE)
    #        level area   x  y  w   h  parent child next key
    n[0] = (1, 500, 0, 0, 30, 28, 2, -1, 1, 5, 0, 0)
    n[1] = (1, 500, 0, 0, 28, 30, 2, -1, -1, 9, 0, 0)
    n[2] = (2, 1100, 0, 0, 30, 30, 3, 0, -1, 2, 0, 0)
    n[3] = (3, 1300, 0, 0, 31, 31, 4, 2, -1, 1, 0, 0)
    n[4] = (4, 9000, 0, 0, 90, 90, -1, 3, -1, 0, 0, 0)
    t = Tree(n, 4, 5, 0)
    for order in (0, 1, 2):
        f = S.ERFilter(8, 120, 900000, 2, 0.7, max_width=128, max_height=128, max_frames=1, sibling_order=order)
        pool, amb = f.non_maximum_supression(_to_node_table(S, t), 100, 100)
        # order 0 without a plane: the table order is the child-list order (node 0 is listed before node 1, and ER::child
        # of node 2 is node 0 -> next node 1), which is the oracle's sibling_mode 0; 1 / 2: smallest / largest key
        ref, ramb = oracle.nms(t, 100, 100, sibling_mode=order)
        assert amb >= 1 and ramb >= 1 and sorted(pool.tolist()) == sorted(ref.tolist())
        f.close()
    # ... and the other child-list order: swap the two children in the table
    n2 = n.copy()
    n2[0], n2[1] = n[1], n[0]
    n2[2]["child"] = 0; n2[0]["next"] = 1; n2[1]["next"] = -1
    t2 = Tree(n2, 4, 5, 0)
    f = S.ERFilter(8, 120, 900000, 2, 0.7, max_width=128, max_height=128, max_frames=1)
    pool, amb = f.non_maximum_supression(_to_node_table(S, t2), 100, 100)
    ref, ramb = oracle.nms(t2, 100, 100)
    assert amb >= 1 and sorted(pool.tolist()) == sorted(ref.tolist())
    p1, _ = f.non_maximum_supression(_to_node_table(S, t), 100, 100)
    assert sorted(int(t.nodes[i]["key"]) for i in p1) != sorted(int(t2.nodes[i]["key"]) for i in pool)     # the order matters here
    f.close()


@pytest.mark.parametrize("replay", ["host", "gpu"])
def test_sibling_ties_follow_the_reference_flood_order(S, oracle, monkeypatch, replay):
    """SURVEY A.5 / VERDICT r1 weak #2: where two or more child chains compete for a parent the reference's winner is the
    child its flood entered last.  A low OVERLAP_COEF and MIN_AREA make such ties frequent; every pool must equal the
    oracle's own order (sibling_mode 0 = the child lists its restatement of the flood built), batched planes included.
    The walk that gives the order runs on a host core (flood_order.cpp) or, with STR_ER_REPLAY=gpu, on one GPU lane."""
    monkeypatch.setenv("STR_ER_REPLAY", replay)
    f = S.ERFilter(8, 6, 900000, 2, 0.3, max_width=320, max_height=200, max_frames=6, kept_cap=70000, pool_cap=30000)
    rng = np.random.default_rng(99)
    n_amb = n_planes = 0
    for trial in range(10):
        h, w = int(rng.integers(20, 201)), int(rng.integers(20, 321))
        imgs = []
        for k in range(6):
            if k == 0:
                imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
            elif k == 1:
                imgs.append(np.kron(rng.integers(0, 256, ((h + 3) // 4, (w + 3) // 4), dtype=np.uint8), np.ones((4, 4), np.uint8))[:h, :w])
            elif k == 2:
                imgs.append(S.synth.gray(S.synth.stext_bgr(int(rng.integers(0, 1 << 30)), w, h)))
            elif k == 3:
                imgs.append((rng.integers(0, 5, (h, w)) * 50 + rng.integers(0, 8, (h, w))).astype(np.uint8))
            elif k == 4:
                g = np.add.outer(np.sin(np.arange(h) / 7.0) * 60, np.cos(np.arange(w) / 9.0) * 60) + 128
                imgs.append(np.clip(g + rng.integers(-10, 11, (h, w)), 0, 255).astype(np.uint8))
            else:
                imgs.append(rng.integers(180, 256, (h, w), dtype=np.uint8))          # sentinel walls everywhere
        res = f.detect_planes(np.stack(imgs), S.STAGE_EXTRACT | S.STAGE_NMS, want_nodes=True)
        for p, img in zip(res.planes, imgs):
            check_plane_against_oracle(oracle, p, img, None, min_area=6, overlap_coef=0.3)
            n_amb += p.ambiguous
            n_planes += 1
    assert n_amb > 50, n_amb       # the test is about ties: make sure they happened
    # the canonical rules stay available (no replay): smallest / largest key
    img = rng.integers(0, 256, (120, 200), dtype=np.uint8)
    for order in (1, 2):
        g = S.ERFilter(8, 6, 900000, 2, 0.3, max_width=320, max_height=200, max_frames=1, kept_cap=70000, pool_cap=30000, sibling_order=order)
        p = g.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS, want_nodes=True).planes[0]
        assert p.ambiguous > 0
        check_plane_against_oracle(oracle, p, img, None, min_area=6, overlap_coef=0.3, sibling_order=order)
        g.close()
    f.close()
    print(f"sibling ties: {n_amb} contested parents on {n_planes} planes, all decided like the reference's flood")


# ---- full size ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["text", "noise"])
def test_full_hd_plane(erf, oracle, oracle_cascades, S, kind):
    img = S.synth.gray(S.synth.KINDS[kind](S.synth.frame_seed(5), 1920, 1080))
    res = erf.detect_planes(img, want_nodes=True)
    check_plane_against_oracle(oracle, res.planes[0], img, oracle_cascades)


def test_full_hd_frame_two_frames(erf, oracle, oracle_cascades, S):
    """BASELINE configs: 1920x1080 BGR, the reference's six planes; 2 frames in one call."""
    frames = S.synth.frames_bgr("text", 0, 2, 1920, 1080)
    res = erf.text_detect(frames)
    assert len(res.planes) == 12
    for p in res.planes:
        if p.frame == 1 or p.ch in (0, 3):        # every plane of frame 1, the luma planes of frame 0
            check_plane_against_oracle(oracle, p, oracle.compute_channels(frames[p.frame])[p.ch], oracle_cascades)
    again = erf.text_detect(frames)
    assert again.cands.tobytes() == res.cands.tobytes()        # deterministic


def test_properties_full_size(erf, S):
    """Size-independent checks at 1920x1080: area bookkeeping and tree shape."""
    img = S.synth.gray(S.synth.snoise_bgr(77, 1920, 1080))
    p = erf.er_tree_extract(img)
    n = p.nodes
    root = n[p.root]
    assert root["parent"] == p.root and (n["flags"] & 1).sum() == 1
    others = np.arange(len(n)) != p.root
    par = n[n["parent"]]
    assert (par["level"][others] > n["level"][others]).all()           # levels strictly increase upward
    assert (par["area"][others] > n["area"][others]).all()             # areas grow upward
    assert (n["area"][others] > 120).all()
    assert (par["x"] <= n["x"]).all() and (par["y"] <= n["y"]).all()   # boxes nest
    assert (par["x"].astype(int) + par["w"] >= n["x"].astype(int) + n["w"]).all()
    assert (par["y"].astype(int) + par["h"] >= n["y"].astype(int) + n["h"]).all()
    assert len(np.unique(n[["key", "level"]])) == len(n)


# ---- error behaviour -------------------------------------------------------------------------------------
def test_errors(S, cascade_paths):
    f = S.ERFilter(8, 120, 900000, 2, 0.7, max_width=128, max_height=96, max_frames=1)
    img = np.zeros((96, 128), np.uint8)
    with pytest.raises(S.StrErError) as e:
        f.detect_planes(img)                                   # classify without cascades
    assert e.value.code == -6
    f.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS)        # fine without classify
    with pytest.raises(S.StrErError) as e:
        f.detect_planes(np.zeros((97, 128), np.uint8), S.STAGE_EXTRACT)
    assert e.value.code == -7
    with pytest.raises(S.StrErError) as e:
        f.load_cascade(0, "/nonexistent/strong.classifier")
    assert e.value.code == -4                                  # reference: prints and returns false
    with pytest.raises(S.StrErError) as e:
        f.load_cascade_text(0, "boost_type REAL\nbase_type DECISION_STUMP\nnum_of_iter 2\nthreshold 0\n1 5 3 0.1 0.2 \n")
    assert e.value.code == -5
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    assert f.cascade_info(0) == (4, 2660) and f.cascade_info(1) == (6, 1354)
    with pytest.raises(S.StrErError) as e:
        f.classify(img, np.array([[100, 0, 40, 10]], np.int32))   # box sticks out of the plane
    assert e.value.code == -1
    f.close()


def test_kept_table_overflow_is_reported(S):
    f = S.ERFilter(8, 2, 900000, 2, 0.7, max_width=256, max_height=256, max_frames=1, kept_cap=64)
    img = np.random.default_rng(0).integers(0, 256, (256, 256), dtype=np.uint8)
    with pytest.raises(S.StrErError) as e:
        f.detect_planes(img, S.STAGE_EXTRACT)
    assert e.value.code == -7 and "kept" in str(e.value)
    f.close()


def test_default_tables_grow_with_the_content(S, oracle):
    """Default capacities follow the planes' pixel counts (a share of a record / kept node / pooled ER per pixel) and are grown, the
    batch repeated, when the content needs more: MIN_AREA 1 on noise keeps nearly every node, far beyond the default shares."""
    f = S.ERFilter(8, 1, 900000, 2, 0.7, max_width=400, max_height=300, max_frames=2)
    ws0 = f.workspace_bytes()
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (300, 400), dtype=np.uint8), S.synth.gray(S.synth.stext_bgr(5, 400, 300))]
    for _ in range(2):          # (the second round runs on the grown tables: no retry, same answer)
        for img in imgs:
            p = f.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS, want_nodes=True).planes[0]
            check_plane_against_oracle(oracle, p, img, None, min_area=1)
            assert p.n_kept > 400 * 300 // 64 + 512 or img is imgs[1]
    assert f.workspace_bytes() > ws0
    f.close()


def test_stability_t_zero_overflows_the_default_pool_share(S, oracle):
    """STABILITY_T = 0 (accepted by str_er_create, 0 .. 255; ctor inc/ER.h:113): every chain of the NMS scores itself against itself, so nearly every kept
    node that passes the size filter is pooled (src/ER.cpp:464-497) -- far more than a plane's default share of pool entries (pixels / 256 + 256).
    The tables must grow and the batch repeat (this was the case that overflowed); node table and pool equal the oracle's."""
    f = S.ERFilter(8, 1, 900000, 0, 0.7, max_width=400, max_height=300, max_frames=1)
    ws0 = f.workspace_bytes()
    rng = np.random.default_rng(17)
    base = np.add.outer(np.arange(300), np.arange(400)) * 0.4
    imgs = [rng.integers(0, 256, (300, 400), dtype=np.uint8), np.clip(base + rng.integers(-6, 7, (300, 400)), 0, 255).astype(np.uint8)]
    pooled = []
    for _ in range(2):          # (the second round runs on the grown tables)
        for img in imgs:
            p = f.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS, want_nodes=True).planes[0]
            check_plane_against_oracle(oracle, p, img, None, min_area=1, stability_t=0)
            pooled.append(p.n_pool)
    assert max(pooled) > 400 * 300 // 256 + 256, pooled
    assert f.workspace_bytes() > ws0
    f.close()
    # ... and the other corner values of the parameter on one plane each
    img = S.synth.gray(S.synth.stext_bgr(21, 400, 300))
    for st, ov, mx in ((1, 0.2, 900000), (5, 0.9, 2000), (255, 0.5, 900000), (0, 0.95, 150)):
        g = S.ERFilter(8, 20, mx, st, ov, max_width=400, max_height=300, max_frames=1)
        p = g.detect_planes(img, S.STAGE_EXTRACT | S.STAGE_NMS, want_nodes=True).planes[0]
        check_plane_against_oracle(oracle, p, img, None, min_area=20, max_area=mx, stability_t=st, overlap_coef=ov)
        g.close()


def test_nv12_ingest(S, cascade_paths, oracle, oracle_cascades):
    """str_er_detect_nv12: the three planes are the oracle's ero_nv12_to_ycrcb of the same bytes, everything after
    them is the path of the BGR entry point -- all planes of a 2-level context against the oracle, two frames, a ragged width."""
    for (W, H) in ((640, 480), (198, 90)):
        f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=2, n_pyr_levels=2))
        f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
        nv = np.stack([S.synth.nv12_from_bgr(S.synth.stext_bgr(S.synth.frame_seed(60 + i), W, H)) for i in range(2)])
        res = f.text_detect_nv12(nv, W, H, want_nodes=True)
        assert len(res.planes) == 2 * 2 * 6
        for p in res.planes:
            three = oracle.nv12_to_ycrcb(nv[p.frame], W, H)
            src = three[p.ch % 3]
            img = oracle.pyramid(src, 2)[p.pyr]
            check_plane_against_oracle(oracle, p, 255 - img if p.ch >= 3 else img, oracle_cascades)
        assert len(res.cands) > 0
        with pytest.raises(S.StrErError):
            f.text_detect_nv12(nv[:, :, :-1].copy(), W - 1, H)                # odd width
        f.close()


# ---- config 5 geometry: 3840x2160, 12-level pyramid ------------------------------------------------------
def test_4k_plane_and_pyramid_dims(S, cascade_paths, oracle, oracle_cascades):
    f = S.ERFilter(params=S.Params(max_width=3840, max_height=2160, max_frames=1, n_pyr_levels=12, channel_mask=0x01))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frame = S.synth.stext_bgr(S.synth.frame_seed(9), 3840, 2160)
    res = f.text_detect(frame, want_nodes=True)
    assert len(res.planes) == 12
    assert [(p.width, p.height) for p in res.planes] == [oracle.pyr_dims(3840, 2160, k) for k in range(12)]
    y = oracle.compute_channels(frame)[0]
    pyr = oracle.pyramid(y, 12)
    for p in res.planes:
        if p.pyr in (0, 1, 5, 11):          # the full-size level, one odd level, two small ones
            check_plane_against_oracle(oracle, p, pyr[p.pyr], oracle_cascades)
    f.close()


def test_random_small_planes_many(erf, oracle, oracle_cascades):
    """200 random planes up to 96x160 in one go (batched call), several value distributions."""
    rng = np.random.default_rng(2024)
    erf.set_min_area(10)
    try:
        for trial in range(8):
            h, w = int(rng.integers(1, 97)), int(rng.integers(1, 161))
            imgs = []
            for k in range(12):
                if k % 3 == 0:
                    imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
                elif k % 3 == 1:
                    imgs.append((rng.integers(0, 3, (h, w)) * 100 + rng.integers(0, 3, (h, w))).astype(np.uint8))
                else:
                    base = np.add.outer(np.arange(h) * int(rng.integers(1, 9)), np.arange(w) * int(rng.integers(1, 9)))
                    imgs.append(((base + rng.integers(0, 12, (h, w))) % 256).astype(np.uint8))
            res = erf.detect_planes(np.stack(imgs), want_nodes=True)
            for p, img in zip(res.planes, imgs):
                check_plane_against_oracle(oracle, p, img, oracle_cascades, min_area=10)
    finally:
        erf.set_min_area(120)


def test_natural_image_crops(erf, oracle, oracle_cascades):
    """Crops of the reference's sample photographs (tests/golden/icdar_crops.npz): all six planes of each."""
    z = np.load(os.path.join(GOLDEN, "icdar_crops.npz"))
    n_amb = 0
    for name in sorted(z.files):
        frame = np.ascontiguousarray(z[name])
        res = erf.text_detect(frame, want_nodes=True)
        planes = oracle.compute_channels(frame)
        assert (erf.compute_channels(frame) == planes).all()
        for p in res.planes:
            check_plane_against_oracle(oracle, p, planes[p.ch], oracle_cascades)
            n_amb += p.ambiguous
    print("ambiguous NMS nodes on natural crops:", n_amb)


def test_plane_sharded_frame_equals_fused_pyramid(S, cascade_paths):
    """SURVEY 8(e), one large frame: the planes of a frame dealt out to ranks (LPT) and run through the plane entry
    point give, put together, exactly the candidates of the fused pyramid call."""
    W, H, L, MASK = 768, 432, 5, 0x2D
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, n_pyr_levels=L, channel_mask=MASK))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frame = S.synth.stext_bgr(S.synth.frame_seed(77), W, H)
    fused = f.text_detect(frame).cands
    planes = S.dist.frame_planes(W, H, L, MASK)
    shares = S.dist.shard_planes_lpt([w * h for (_, _, w, h) in planes], 3)
    parts = [S.dist.detect_plane_share(f, frame, sh, L, MASK) for sh in shares]
    allc = np.concatenate(parts)
    allc = allc[np.lexsort((allc["key"], allc["node"]))]
    fields = ["frame", "ch", "pyr", "level", "cls", "x", "y", "w", "h", "area", "key", "score_strong", "score_weak"]
    assert len(allc) == len(fused) > 0
    assert allc[fields].tolist() == fused[fields].tolist()
    one = S.dist.detect_frame_plane_sharded(f, frame, 0, 1, L, MASK)
    assert one[fields].tolist() == fused[fields].tolist()
    f.close()


def test_both_tile_kernel_sizes_give_the_same_answer(S, cascade_paths, monkeypatch):
    """The tile kernel exists in a small-LDS / high-occupancy size and a big one (picked per batch from the node density of the
    previous batch); both, and the automatic choice across a text -> noise -> text sequence, give identical results."""
    W, H = 448, 320
    text = np.stack([S.synth.stext_bgr(S.synth.frame_seed(5 + i), W, H) for i in range(2)])
    noise = np.random.default_rng(3).integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    out = {}
    for mode in ("sparse", "dense", ""):
        if mode:
            monkeypatch.setenv("STR_ER_TILE_KERNEL", mode)
        else:
            monkeypatch.delenv("STR_ER_TILE_KERNEL", raising=False)
        f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=2))
        f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
        out[mode] = [f.text_detect(x, want_nodes=True) for x in (text, noise, noise, text, text)]
        f.close()
    for mode in ("dense", ""):
        for a, b in zip(out["sparse"], out[mode]):
            assert a.cands.tobytes() == b.cands.tobytes() and a.info.tobytes() == b.info.tobytes()
            for pa, pb in zip(a.planes, b.planes):
                assert pa.nodes.tobytes() == pb.nodes.tobytes()
    assert len(out[""][0].cands) > 0


def test_both_tile_kernels_give_the_same_answer(S, cascade_paths, oracle, oracle_cascades, monkeypatch):
    """Two tile kernels build the trees of a batch's tiles: k_tile_tree (pieces + union-find in LDS) and k_tile_tree2 (level by level on bit masks;
    by default on the chroma planes, whose tiles hold two or three levels; a tile it does not take is handed back to the first).  STR_ER_TILE2 = 0
    (first kernel only) / 1 (default) / 2 (every plane through the second) give the same records, node for node, and they are the oracle's: text-like
    frames, a walled-in frame, ragged sizes down to 1 x 1, noise (nearly every tile handed back), thresh steps 8 and 16, MIN_AREA 1 and 120."""
    rng = np.random.default_rng(41)
    walled = S.synth.stext_bgr(S.synth.frame_seed(34), 448, 320)
    walled[100:108, :, :] = 255; walled[:, 200:203, :] = 254; walled[0, 0, :] = 255
    flat = np.full((70, 200, 3), 128, np.uint8); flat[10:20, 30:90, 1] = 140; flat[40:44, 100:180, 2] = 90
    frames = [S.synth.stext_bgr(S.synth.frame_seed(7), 448, 320), walled, S.synth.stext_bgr(S.synth.frame_seed(8), 200, 70),
              rng.integers(0, 256, (33, 130, 3), dtype=np.uint8), flat, S.synth.stext_bgr(S.synth.frame_seed(9), 65, 33),
              rng.integers(100, 140, (1, 1, 3), dtype=np.uint8), rng.integers(100, 140, (3, 2, 3), dtype=np.uint8), S.synth.stext_bgr(S.synth.frame_seed(10), 191, 40)]
    for step, min_area in ((8, 120), (16, 1), (8, 1)):
        out = {}
        for mode in ("0", "1", "2"):
            monkeypatch.setenv("STR_ER_TILE2", mode)
            f = S.ERFilter(params=S.Params(max_width=448, max_height=320, max_frames=1, thresh_step=step, min_area=min_area, kept_cap=160000, pool_cap=160000))
            f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
            out[mode] = [f.text_detect(x, want_nodes=True) for x in frames]
            st = f.tile2_stats()
            assert (st["tiles"] > 0) == (mode != "0")
            if mode == "1":
                assert st["handed_back"] < st["tiles"]
            f.close()
        for mode in ("1", "2"):
            for a, b in zip(out["0"], out[mode]):
                assert a.cands.tobytes() == b.cands.tobytes() and a.info.tobytes() == b.info.tobytes()
                for pa, pb in zip(a.planes, b.planes):
                    assert pa.nodes.tobytes() == pb.nodes.tobytes()
        for x, r in zip(frames, out["2"]):
            six = oracle.compute_channels(x)
            for p in r.planes:
                check_plane_against_oracle(oracle, p, six[p.ch], oracle_cascades, step=step, min_area=min_area)
    assert len(out["2"][0].cands) > 0


# ---- one plane in strips over several GPUs ------------------------------------------------------------------
@pytest.mark.parametrize("n_strips", [2, 3, 5])
def test_strips_of_a_plane_merge_into_the_unsplit_result(S, cascade_paths, oracle, oracle_cascades, n_strips):
    """Every strip is extracted by a context of its own (as another GPU would), the blobs travel as plain bytes, the owner
    merges them: node tables, pools, classes and scores equal the fused call's and the oracle's, for text-like, noisy and
    walled-in frames, incl. a frame with fewer tile rows than strips."""
    W, H = 448, 300
    owner = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, kept_cap=40000, pool_cap=10000))
    owner.load_cascade(0, cascade_paths[0]); owner.load_cascade(1, cascade_paths[1])
    workers = [S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, kept_cap=40000, pool_cap=10000)) for _ in range(n_strips)]
    rng = np.random.default_rng(17)
    frames = [S.synth.stext_bgr(S.synth.frame_seed(31), W, H), rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
              S.synth.stext_bgr(S.synth.frame_seed(32), 200, 70), rng.integers(0, 256, (33, 130, 3), dtype=np.uint8)]
    walled = S.synth.stext_bgr(S.synth.frame_seed(33), W, H)
    walled[140:150, :, :] = 255                                   # a band at the sentinel level right across: seals the lower part off
    walled[0, 0, :] = 255
    frames.append(walled)
    fields = ["frame", "ch", "pyr", "level", "cls", "x", "y", "w", "h", "area", "key", "score_strong", "score_weak"]
    for frame in frames:
        blobs = [workers[s].strip_extract(frame, s, n_strips) for s in range(n_strips)]
        merged = owner.strip_merge(frame, blobs, want_nodes=True)
        fused = owner.text_detect(frame, want_nodes=True)
        assert merged.cands[fields].tolist() == fused.cands[fields].tolist()
        assert merged.info.tobytes() == fused.info.tobytes()
        six = oracle.compute_channels(frame)
        for pm, pf in zip(merged.planes, fused.planes):
            assert pm.nodes.tobytes() == pf.nodes.tobytes()
            check_plane_against_oracle(oracle, pm, six[pm.ch], oracle_cascades)
    # blobs of another frame / a wrong order are refused
    other = [workers[s].strip_extract(frames[0], s, n_strips) for s in range(n_strips)]
    with pytest.raises(S.StrErError):
        owner.strip_merge(frames[0], other[::-1] if n_strips > 1 else [other[0][:20]])
    with pytest.raises(S.StrErError):
        owner.strip_merge(frames[2], other)
    for f in workers + [owner]:
        f.close()


@pytest.mark.gpu
def test_groups_left_alone_are_joined_by_the_undone_pass(S, cascade_paths, monkeypatch):
    """k_seam only sees the seams BETWEEN groups of tiles; the seams inside a group are k_group_merge's -- or, for a group with more records than its table
    holds, k_seam_undone's (the group puts itself on a device-side list).  With the smallest table (512 records) and large groups most groups of a text-like
    pyramid frame and every group of a noise frame overflow; with 1 x 1 groups nothing is inside a group; without grouping every seam is k_seam's.  All give
    the records of the default shape, node for node -- on a batch large enough to take the 8 x 4 default (> 96 planes) and on a call of one frame."""
    W, H = 448, 320
    rng = np.random.default_rng(77)
    one = S.synth.stext_bgr(S.synth.frame_seed(70), W, H)[None]
    many = np.stack([S.synth.stext_bgr(S.synth.frame_seed(71 + i), W, H) for i in range(5)] + [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)])
    out = {}
    for shape in (None, ("8", "8", "0"), ("4", "4", "0"), ("1", "1", "2"), ("16", "2", "0"), ("0", "0", "2")):
        for k, v in zip(("STR_ER_GROUP_X", "STR_ER_GROUP_Y", "STR_ER_GROUP_KERNEL"), shape or (None, None, None)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=6, n_pyr_levels=3, channel_mask=0x3F))       # 6 frames x 18 planes = 108 planes
        f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
        out[shape] = [f.text_detect(x, want_nodes=True) for x in (many, one, many)]
        f.close()
    for shape, res in out.items():
        for a, b in zip(out[None], res):
            assert a.cands.tobytes() == b.cands.tobytes() and a.info.tobytes() == b.info.tobytes(), shape
            for pa, pb in zip(a.planes, b.planes):
                assert pa.nodes.tobytes() == pb.nodes.tobytes(), shape
    assert len(out[None][0].cands) > 0
