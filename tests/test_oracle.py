"""CPU tests of the parity oracle (oracle/er_oracle.c): known answers, an independent
brute-force labelling, and the reference's own cascade code (oracle/_ref)."""
import os

import numpy as np
import pytest

from conftest import oracle_tree_canon

DBL_MAX = float(np.finfo(np.float64).max)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def root(t):
    return t.nodes[t.root]


# ---- quantisation (src/ER.cpp:247-250) ---------------------------------------------
def test_quant_lut_half_to_even(oracle):
    lut = oracle.quant_lut(8)
    assert [int(lut[v]) for v in (0, 3, 4, 5, 12, 20, 28, 100, 251, 252, 255)] == [0, 0, 0, 1, 2, 2, 4, 12, 31, 32, 32]
    assert oracle.highest_level(8) == 32 and oracle.highest_level(1) == 256 and oracle.highest_level(2) == 128
    assert int(oracle.quant_lut(2)[255]) == 128          # 127.5 rounds to even -> the sentinel level
    assert (oracle.quant_lut(1) == np.arange(256)).all()


# ---- known answers recorded in SURVEY.md Appendix B (runs of the unmodified reference) ---------
@pytest.mark.parametrize("value,level", [(100, 12), (20, 2), (28, 4)])
def test_constant_plane(oracle, value, level):
    t = oracle.tree_extract(np.full((10, 10), value, np.uint8), 8, 0)
    r = root(t)
    assert len(t.nodes) == 1 and r["level"] == level and r["area"] == 101 and (r["w"], r["h"]) == (10, 10)


def test_constant_sentinel_plane(oracle):
    t = oracle.tree_extract(np.full((10, 10), 252, np.uint8), 8, 0)
    r = root(t)
    assert (r["level"], r["area"], r["x"], r["y"], r["w"], r["h"]) == (32, 2, 0, 0, 1, 1)


def test_bright_ring_hides_interior(oracle):
    img = np.full((9, 9), 40, np.uint8)
    img[1, 1:8] = img[7, 1:8] = 255
    img[1:8, 1] = img[1:8, 7] = 255
    r = root(oracle.tree_extract(img, 8, 0))
    assert r["area"] == 32 + 1 and r["npix"] == 32      # the 25 interior pixels are never flooded


def test_start_rules(oracle):
    m = np.full((6, 6), 40, np.uint8)
    m[0, 0] = 255
    r = root(oracle.tree_extract(m, 8, 0))
    assert (r["area"], r["w"], r["h"]) == (36, 6, 6)     # 35 px + 1, entered through pixel 1
    m[0, 1] = 255
    assert root(oracle.tree_extract(m, 8, 0))["area"] == 35  # through pixel w
    m[1, 0] = 255
    r = root(oracle.tree_extract(m, 8, 0))
    assert (r["level"], r["area"], r["w"], r["h"]) == (32, 2, 1, 1)


def test_single_pixel_and_line(oracle):
    r = root(oracle.tree_extract(np.array([[77]], np.uint8), 8, 0))
    assert (r["level"], r["area"]) == (10, 2)
    t = oracle.tree_extract(np.arange(0, 248, 8, dtype=np.uint8)[None, :], 8, 0)   # 1x31 ramp: one node per pixel
    assert len(t.nodes) == 31 and root(t)["area"] == 31 + 31


# ---- flood == canonical node set ---------------------------------------------------
def _planes(rng):
    for trial in range(48):
        h, w = int(rng.integers(1, 48)), int(rng.integers(1, 48))
        mode = trial % 4
        if mode == 0:
            yield rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif mode == 1:
            yield (rng.integers(0, 4, (h, w)) * 60).astype(np.uint8)
        elif mode == 2:
            yield rng.integers(200, 256, (h, w)).astype(np.uint8)
        else:
            yield (np.add.outer(np.arange(h) * 3, np.arange(w) * 5) % 256).astype(np.uint8)


def test_flood_matches_bruteforce(oracle):
    rng = np.random.default_rng(0)
    for img in _planes(rng):
        for step, ma in ((8, 0), (8, 5), (2, 3), (1, 0), (16, 2)):
            a, b = oracle.tree_extract(img, step, ma), oracle.tree_bruteforce(img, step, ma)
            assert oracle_tree_canon(a) == oracle_tree_canon(b)
            assert a.dead_branch == 0          # src/ER.cpp:169-178 never runs


def test_area_is_pixels_plus_subtree_nodes(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    t = oracle.tree_extract(img, 8, 0)
    assert (t.nodes["area"] == t.nodes["npix"] + t.nodes["nsub"]).all()
    r = root(t)
    assert r["nsub"] == len(t.nodes)


# ---- NMS (src/ER.cpp:416-505) --------------------------------------------------------------------
def test_nms_order_independent_when_unambiguous(oracle, S):
    img = S.synth.gray(S.synth.stext_bgr(11, 320, 240))
    t = oracle.tree_extract(img, 8, 60)
    p0, amb = oracle.nms(t, 240, 320, min_area=60)
    p1, _ = oracle.nms(t, 240, 320, min_area=60, sibling_mode=1)
    p2, _ = oracle.nms(t, 240, 320, min_area=60, sibling_mode=2)
    assert len(p0) > 0
    if amb == 0:
        assert sorted(p0) == sorted(p1) == sorted(p2)


def test_nms_chain_rules(oracle):
    """Hand-made chain: nested boxes 20x20 < 21x21 < 22x22 < 40x40; T=2, coef 0.7."""
    from oracle.oracle import NODE_DTYPE, Tree
    n = np.zeros(4, NODE_DTYPE)
    for i, (w, lvl, area) in enumerate([(20, 1, 500), (21, 2, 600), (22, 3, 700), (40, 4, 2000)]):
        n[i] = (lvl, area, 0, 0, w, w, i + 1 if i < 3 else -1, i - 1 if i > 0 else -1, -1, i, 0, 0)
    t = Tree(n, 3, 4, 0)
    pool, amb = oracle.nms(t, 100, 100)
    # chain {0,1,2} (400/441, 400/484 > 0.7; 400/1600 is not): stab0 = 400/(484-400); only i=0 is eligible (len-T = 1) -> node 0; node 3 alone: no pool
    assert list(pool) == [0] and amb == 0


# ---- classify chain -------------------------------------------------------------------------------
def test_resize_identities(oracle):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (26, 26), dtype=np.uint8)
    assert (oracle.resize(a, 26, 26) == a).all()
    b = rng.integers(0, 256, (52, 48), dtype=np.uint8)
    area = ((b[0::2, 0::2].astype(int) + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert (oracle.resize(b, 24, 26) == area).all()      # exact 2x in both axes -> INTER_AREA path
    c = np.full((37, 91), 173, np.uint8)
    assert (oracle.resize(c, 26, 13) == 173).all()       # bilinear of a constant is the constant


def test_aran_placement(oracle):
    roi = np.full((20, 80), 200, np.uint8)                # wide: 26 x int(26*sqrt(0.25)) = 26 x 13, rows 6..18
    t = oracle.aran26(roi)
    assert (t[6:19, :] == 200).all() and (t[:6] == 0).all() and (t[19:] == 0).all()
    roi = np.full((80, 20), 200, np.uint8)                # tall: 13 x 26, columns 6..18
    t = oracle.aran26(roi)
    assert (t[:, 6:19] == 200).all() and (t[:, :6] == 0).all() and (t[:, 19:] == 0).all()


def test_aran_size_pow_equals_sqrt(oracle):
    """The kernel computes (int)(26*sqrt(R1)); the reference (int)(26*pow(R1,0.5)) (src/OCR.cpp:397).
    With this libm the two agree for every ROI size up to 2200x2200 (exact squares like 676x441 included)."""
    assert oracle.lib.ero_selftest_pow_vs_sqrt(2200) == 0
    assert oracle.aran_dims(676, 441) == (26, 21) and oracle.aran_dims(441, 676) == (21, 26)
    assert oracle.aran_dims(52, 52) == (26, 26) and oracle.aran_dims(48, 52) == (24, 26)


def test_lbp_hist_shape(oracle):
    rng = np.random.default_rng(3)
    roi = rng.integers(0, 256, (33, 47), dtype=np.uint8)
    hist = oracle.lbp_hist(roi)
    assert hist.sum() == 576 and [hist[i * 256:(i + 1) * 256].sum() for i in range(4)] == [144] * 4
    flat = oracle.lbp_hist(np.full((30, 30), 90, np.uint8))
    assert flat[0] == 144 and flat[256] == 144           # nothing exceeds the neighbourhood mean -> code 0


def test_lbp_stride_quirk(oracle):
    """calc_LBP addresses the 26-wide tile with stride 24 (src/ER.cpp:828-840)."""
    rng = np.random.default_rng(4)
    tile = rng.integers(0, 256, (26, 26), dtype=np.uint8)
    lbp = oracle.lbp24(tile)
    flat = tile.reshape(-1).astype(int)
    offs = [-25, -24, -23, 1, 25, 24, 23, -1]
    for i, j in ((0, 0), (5, 7), (23, 23), (12, 0)):
        c = (i + 1) * 26 + (j + 1)
        v = [flat[c + o] for o in offs]
        code = sum((8 * x > sum(v)) << k for k, x in enumerate(v))
        assert lbp[i, j] == code


def test_channels_known_colours(oracle):
    bgr = np.array([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 90]]], np.uint8)
    p = oracle.compute_channels(bgr)
    assert p[0, 0].tolist() == [0, 255, 29, 150, 76, 145]            # Y
    assert p[1, 0, :2].tolist() == [128, 128] and p[2, 0, :2].tolist() == [128, 128]
    assert p[1, 0, 4] == 255 and p[2, 0, 2] == 255                    # saturated Cr of red, Cb of blue
    assert (p[3:] == 255 - p[:3]).all()


# ---- cascade: pinned against the reference's own code -----------------------------------------------
def test_cascade_loader_counts(oracle_cascades):
    s, w = oracle_cascades
    assert (s.n_stages, s.n_stumps) == (4, 2660) and (w.n_stages, w.n_stumps) == (6, 1354)


def test_cascade_matches_reference_vectors(oracle_cascades):
    """tests/golden/cascade_vectors.npz holds outputs of the REAL CascadeBoost::predict."""
    z = np.load(os.path.join(GOLDEN, "cascade_vectors.npz"))
    s, w = oracle_cascades
    for h, es, ew in zip(z["hist"], z["strong"], z["weak"]):
        fv = h.astype(np.float64)
        assert s.predict(fv) == es and w.predict(fv) == ew


def test_cascade_matches_reference_library(oracle_cascades, cascade_paths):
    from oracle.oracle import RefCascade
    if not RefCascade.available():
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt library)")
    rs, rw = RefCascade(cascade_paths[0]), RefCascade(cascade_paths[1])
    assert rs.n_stumps == 2660 and rw.n_stumps == 1354
    rng = np.random.default_rng(5)
    s, w = oracle_cascades
    for _ in range(200):
        fv = np.concatenate([np.bincount(rng.integers(0, 256, 144) // rng.integers(1, 40), minlength=256) for _ in range(4)])
        fv = fv.astype(np.float64)
        assert s.predict(fv) == rs.predict(fv) and w.predict(fv) == rw.predict(fv)


def test_golden_text_round_trip(S):
    """cascade_io rebuilds the reference's text format from the fixture without losing a bit."""
    z = np.load(S.cascade_io.GOLDEN)
    txt = S.cascade_io.golden_text("strong")
    rows = [l.split() for l in txt.split("\n")[4:] if l.strip()]
    assert len(rows) == 2660
    assert all(l.endswith(" ") for l in txt.split("\n")[4:-1])
    assert np.array_equal(np.array([float(r[3]) for r in rows]), z["strong_cp"])
    assert np.array_equal(np.array([float(r[4]) for r in rows]), z["strong_cn"])


def test_pyramid_dims(oracle):
    dims = [oracle.pyr_dims(1920, 1080, k) for k in range(8)]
    assert dims == [(1920, 1080), (1358, 764), (960, 540), (679, 382), (480, 270), (339, 191), (240, 135), (170, 95)]


def test_nv12_ingest_definition(oracle):
    """The build-defined NV12 -> Y/Cr/Cb step (no reference counterpart; oracle/er_oracle.c: ero_nv12_to_ycrcb): luma as is, Cb / Cr
    = the decoder's U / V sample of the pixel's 2 x 2 block."""
    w, h = 6, 4
    nv = np.arange((h + h // 2) * w, dtype=np.uint8).reshape(h + h // 2, w)
    pl = oracle.nv12_to_ycrcb(nv, w, h)
    assert (pl[0] == nv[:h]).all()
    for y in range(h):
        for x in range(w):
            assert pl[2][y, x] == nv[h + y // 2, 2 * (x // 2)]          # Cb <- U
            assert pl[1][y, x] == nv[h + y // 2, 2 * (x // 2) + 1]      # Cr <- V
