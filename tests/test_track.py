"""calc_color + er_track (src/ER.cpp:530-590, 1391-1419).

CPU tests pin the oracle on hand-made cases; GPU tests compare the HIP path with the oracle on the
candidates of synthetic frames (bit-exact: the colours are quotients of integer sums, the rest is integer)."""
import numpy as np
import pytest


def _ers(oracle, rows):
    e = np.zeros(len(rows), oracle.ER_DTYPE)
    for i, r in enumerate(rows):
        for k, v in r.items():
            e[i][k] = v
        e[i]["id"] = i
    return e


def test_oracle_calc_color_reads_image_origin(oracle):
    """calc_color masks the box on its own channel but reads the colour image from row 0 / column 0 (src/ER.cpp:1404)."""
    mask = np.full((40, 60), 200, np.uint8)
    mask[10:20, 30:40] = 30                       # dark glyph in the box -> 255 - v is bright -> inside the Otsu mask
    box = (25, 5, 20, 20)
    col = np.zeros((40, 60, 3), np.uint8)
    col[..., 0] = np.arange(60)[None, :]         # channel 0 = column index
    col[..., 1] = np.arange(40)[:, None]         # channel 1 = row index
    col[..., 2] = 7
    c = oracle.calc_color(mask, col, box)
    # the glyph sits at box-relative columns 5..14, rows 5..14 -> the colour image is sampled THERE, not at 30..39 / 10..19
    assert c[0] == np.mean(np.arange(5, 15)) and c[1] == np.mean(np.arange(5, 15)) and c[2] == 7.0
    flat = np.full((40, 60), 90, np.uint8)        # constant box: Otsu threshold 0 and 255-90 > 0 -> everything is inside the mask
    c = oracle.calc_color(flat, col, box)
    assert c[2] == 7.0 and c[0] == np.mean(np.arange(20))
    white = np.full((40, 60), 255, np.uint8)      # 255 - 255 = 0 is never > threshold: empty mask -> 0.0 / 0
    assert np.isnan(oracle.calc_color(white, col, box)).all()


def test_oracle_er_track_closure(oracle):
    """all_er = strong ERs, then every weak ER tied to something already in all_er -- transitively (src/ER.cpp:563-590)."""
    base = dict(y=100, w=20, h=30, area=400, color1=100.0, color2=120.0, color3=130.0)
    rows = [dict(base, x=100, cls=1),                       # 0 strong
            dict(base, x=150, cls=2),                       # 1 weak, near 0 (|dx| = 50 < 2*30)
            dict(base, x=205, cls=2),                       # 2 weak, too far from 0 (105) but near 1 (55): joins through 1
            dict(base, x=400, cls=2),                       # 3 weak, far from everything
            dict(base, x=120, cls=2, color1=130.0),         # 4 weak, colour differs by 30
            dict(base, x=125, cls=2, h=61),                 # 5 weak, |dh| = 31 >= min(h) = 30
            dict(base, x=130, cls=2, area=1600),            # 6 weak, |da| = 1200 >= 3 * 400
            dict(base, x=135, cls=0),                       # 7 pool-only: ignored
            dict(base, x=140, cls=2, color2=float("nan"))]  # 8 weak, NaN colour never matches
    order, e = oracle.er_track(_ers(oracle, rows))
    assert list(order) == [0, 1, 2]
    assert (e["cx"] == e["x"] + e["w"] // 2).all() and (e["cy"] == e["y"] + e["h"] // 2).all()
    # the rule is evaluated from the tracked ER's side: max(s.w, s.h) << 1 uses s only
    rows = [dict(base, x=100, cls=1, w=10, h=10, area=100), dict(base, x=123, cls=2, w=14, h=14, area=150)]
    order, _ = oracle.er_track(_ers(oracle, rows))
    assert list(order) == [0]                               # |dcx| + |dcy| = 25 + 2 >= 20
    rows[0]["cls"], rows[1]["cls"] = 2, 1
    order, _ = oracle.er_track(_ers(oracle, rows))
    assert list(order) == [1, 0]                            # seen from the 14x14 one: 27 < 28


def _ycrcb(oracle, bgr):
    h, w = bgr.shape[:2]
    planes = np.zeros((6, h, w), np.uint8)
    import ctypes as C
    p = C.POINTER(C.c_uint8)
    oracle.lib.ero_compute_channels(np.ascontiguousarray(bgr).ctypes.data_as(p), w * 3, w, h, planes.ctypes.data_as(p))
    return planes


@pytest.mark.gpu
def test_gpu_calc_color_matches_oracle(erf, oracle, S):
    bgr = S.synth.stext_bgr(S.synth.frame_seed(2), 320, 240)
    planes = _ycrcb(oracle, bgr)
    col = np.ascontiguousarray(np.stack([planes[0], planes[1], planes[2]], axis=-1))
    rng = np.random.default_rng(9)
    boxes = []
    for _ in range(120):
        bw, bh = int(rng.integers(1, 200)), int(rng.integers(1, 160))
        boxes.append((int(rng.integers(0, 320 - bw + 1)), int(rng.integers(0, 240 - bh + 1)), bw, bh))
    boxes += [(0, 0, 320, 240), (319, 239, 1, 1)]
    boxes = np.array(boxes, np.int32)
    for mask in (planes[0], planes[4], np.full((240, 320), 255, np.uint8)):
        got = erf.calc_color(mask, col, boxes)
        for b, g in zip(boxes, got):
            exp = oracle.calc_color(mask, col, b)
            assert np.array_equal(g, exp, equal_nan=True), (b, g, exp)
    with pytest.raises(S.StrErError):
        erf.calc_color(planes[0], col[:100], np.array([[0, 0, 10, 150]], np.int32))      # box taller than the colour image


def _expected_tracks(oracle, res, planes_of_frame):
    """Oracle calc_color + er_track over the candidates of one image of a detect result."""
    exp = []
    for (f, pyr), pl in planes_of_frame.items():
        sel = np.nonzero((res.cands["frame"] == f) & (res.cands["pyr"] == pyr))[0]
        col = np.ascontiguousarray(np.stack([pl[0], pl[1], pl[2]], axis=-1))
        ers = np.zeros(len(sel), oracle.ER_DTYPE)
        for k, i in enumerate(sel):
            c = res.cands[i]
            ers[k]["x"], ers[k]["y"], ers[k]["w"], ers[k]["h"] = c["x"], c["y"], c["w"], c["h"]
            ers[k]["area"], ers[k]["cls"], ers[k]["ch"], ers[k]["id"] = c["area"], c["cls"], c["ch"], i
            if c["cls"]:
                ers[k]["color1"], ers[k]["color2"], ers[k]["color3"] = oracle.calc_color(pl[c["ch"]], col, (c["x"], c["y"], c["w"], c["h"]))
        order, e = oracle.er_track(ers)
        exp.append((sel, order, e))
    return exp


@pytest.mark.gpu
def test_gpu_track_stage_matches_oracle(S, cascade_paths, oracle):
    """text_detect through er_track on the reference's native 6 planes: colours, centres and the tracked set."""
    W, H, F = 480, 360, 3
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frames = np.stack([S.synth.stext_bgr(S.synth.frame_seed(20 + i), W, H) for i in range(F)])
    res = f.text_detect(frames, S.STAGE_ALL | S.STAGE_TRACK)
    assert res.tracks is not None and len(res.tracks) == len(res.cands)
    assert res.times[3] > 0
    planes = {(i, 0): _ycrcb(oracle, frames[i]) for i in range(F)}
    n_tracked = 0
    for sel, order, e in _expected_tracks(oracle, res, planes):
        t = res.tracks[sel]
        live = e["cls"] != 0
        assert np.array_equal(t["color1"][live], e["color1"][live], equal_nan=True)
        assert np.array_equal(t["color2"][live], e["color2"][live], equal_nan=True)
        assert np.array_equal(t["color3"][live], e["color3"][live], equal_nan=True)
        assert (t["cx"][live] == e["cx"][live]).all() and (t["cy"][live] == e["cy"][live]).all()
        want = np.zeros(len(sel), bool)
        want[order] = True
        assert (t["tracked"].astype(bool) == want).all()
        assert (t["tracked"][~live] == 0).all()
        n_tracked += int(want.sum())
    assert n_tracked > 0
    # the single-stage entry point gives the same set
    sel, order, e = _expected_tracks(oracle, res, planes)[0]
    cols = np.stack([res.tracks[sel]["color1"], res.tracks[sel]["color2"], res.tracks[sel]["color3"]], axis=1)
    tr, cx, cy = f.er_track(res.cands[sel], cols)
    want = np.zeros(len(sel), bool)
    want[order] = True
    assert (tr == want).all() and (cx == e["cx"]).all()
    f.close()


@pytest.mark.gpu
def test_gpu_track_stage_pyramid_and_errors(S, cascade_paths, oracle):
    """With pyramid levels an image is one frame at one level; gray planes have no colour image."""
    W, H = 384, 256
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=2, n_pyr_levels=3, channel_mask=0x2D))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frames = np.stack([S.synth.stext_bgr(S.synth.frame_seed(31 + i), W, H) for i in range(2)])
    res = f.text_detect(frames, S.STAGE_ALL | S.STAGE_TRACK)
    planes = {}
    for i in range(2):
        lvl = _ycrcb(oracle, frames[i])[:3]
        for l in range(3):
            if l > 0:
                dw, dh = oracle.pyr_dims(W, H, l)
                lvl = np.stack([oracle.resize(p, dw, dh) for p in lvl])
            full = np.concatenate([lvl, 255 - lvl])
            planes[(i, l)] = full
    for sel, order, e in _expected_tracks(oracle, res, planes):
        t = res.tracks[sel]
        live = e["cls"] != 0
        for k in ("color1", "color2", "color3"):
            assert np.array_equal(t[k][live], e[k][live], equal_nan=True)
        want = np.zeros(len(sel), bool)
        want[order] = True
        assert (t["tracked"].astype(bool) == want).all()
    with pytest.raises(S.StrErError):
        f.detect_planes(S.synth.gray(frames[0]), S.STAGE_ALL | S.STAGE_TRACK)
    with pytest.raises(S.StrErError):
        f.text_detect(frames, S.STAGE_EXTRACT | S.STAGE_NMS | S.STAGE_TRACK)
    f.close()
