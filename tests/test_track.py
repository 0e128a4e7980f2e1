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


# ---- er_grouping (src/ER.cpp:612-692) -------------------------------------------------
def test_oracle_fitline_avgslope(oracle):
    assert oracle.fitline_avgslope([(0, 0), (10, 1)]) == 0.0                               # <= 2 points (:1363)
    assert oracle.fitline_avgslope([(0, 0), (10, 1), (20, 2), (30, 3)]) == pytest.approx(0.1)
    # one outlier in a triple: the smallest |slope| of the three is taken (:1375-1380)
    s = oracle.fitline_avgslope([(0, 0), (10, 0), (20, 9)])
    assert s == 0.0
    # a vertical pair gives +-inf, which never passes a `<` test against the finite ones but poisons nothing
    assert oracle.fitline_avgslope([(0, 0), (0, 5), (10, 5)]) == 0.0


def test_oracle_er_grouping_lines(oracle):
    """Greedy line assignment in sorted order; a line never merges with another one (:650-661)."""
    base = dict(y=100, w=20, h=30, area=400, color1=100.0, color2=120.0, color3=130.0, cls=1)
    rows = [dict(base, x=300), dict(base, x=100), dict(base, x=130), dict(base, x=160, y=101), dict(base, x=330, y=140),
            dict(base, x=600, color3=200.0)]
    e = _ers(oracle, rows)
    e["cx"], e["cy"] = e["x"] + e["w"] // 2, e["y"] + e["h"] // 2
    all_idx, lines, after = oracle.er_grouping(e)
    assert list(all_idx) == [1, 2, 3, 0, 4, 5]                                              # sorted by center.x
    assert [list(l[0]) for l in lines] == [[1, 2, 3]]                                       # 0 and 4 differ by 40 in y: |dy| >= (30+30)/4
    assert lines[0][2] == (100, 100, 80, 31)
    assert lines[0][1] == pytest.approx((0.0 + 1 / 30 + 1 / 60) / 3)                        # slopes 0, 1/30, 1/60 are within epsilon: their mean
    # equal center.x: the tie keeps the input order (stable sort)
    rows = [dict(base, x=100, y=100), dict(base, x=100, y=104, w=20), dict(base, x=70)]
    e = _ers(oracle, rows)
    e["cx"], e["cy"] = e["x"] + e["w"] // 2, e["y"] + e["h"] // 2
    all_idx, lines, after = oracle.er_grouping(e)
    assert list(all_idx) == [2, 0, 1]
    # members 0 and 1 overlap by > 0.5: overlap_suppression merges 1 into 0 (averaged bound) while computing the slope (:670-672)
    assert (after[0]["y"], after[0]["h"], after[0]["cy"]) == (102, 30, 117) and after[1]["y"] == 104
    # pair (0, 1) finds both already in the line and pushes 0 AGAIN (:650-654): members repeat, as in the reference
    assert [list(l[0]) for l in lines] == [[2, 0, 1, 0]]
    # inner_sup: an ER sitting concentrically inside a more than twice larger one goes before the pairing (:893-922)
    rows = [dict(base, x=100, w=40, h=40, area=900), dict(base, x=110, y=110, w=20, h=20, area=300), dict(base, x=150, w=36, h=40, area=800)]
    e = _ers(oracle, rows)
    e["cx"], e["cy"] = e["x"] + e["w"] // 2, e["y"] + e["h"] // 2
    assert list(oracle.er_grouping(e, inner_sup=True)[0]) == [0, 2]
    assert list(oracle.er_grouping(e, inner_sup=False)[0]) == [0, 1, 2]


def _expected_lines(oracle, res, sel, inner_sup, overlap_sup=False):
    """Oracle er_grouping on the tracked candidates `sel` (indices into res.cands, candidate order)."""
    tr = res.tracks[sel]
    keep = sel[tr["tracked"] != 0]
    e = np.zeros(len(keep), oracle.ER_DTYPE)
    for k, i in enumerate(keep):
        c, t = res.cands[i], res.tracks[i]
        e[k]["x"], e[k]["y"], e[k]["w"], e[k]["h"], e[k]["area"] = c["x"], c["y"], c["w"], c["h"], c["area"]
        e[k]["cx"], e[k]["cy"] = t["cx"], t["cy"]
        e[k]["color1"], e[k]["color2"], e[k]["color3"] = t["color1"], t["color2"], t["color3"]
        e[k]["id"] = i
    all_idx, lines, after = oracle.er_grouping(e, overlap_sup=overlap_sup, inner_sup=inner_sup)
    return keep, lines, after, keep[all_idx]


def _check_lines(res, groups, oracle, inner_sup, overlap_sup=False):
    """groups: list of candidate-index arrays, one per image, in image order."""
    li = 0
    n_lines = 0
    all_er = []
    for sel in groups:
        keep, lines, after, ga = _expected_lines(oracle, res, sel, inner_sup, overlap_sup)
        all_er += [int(v) for v in ga]
        for members, slope, box in lines:
            t = res.texts[li]
            got = res.text_ers[t["first"]:t["first"] + t["count"]]
            assert list(got) == [int(keep[m]) for m in members], li
            assert (t["slope"] == slope) or (np.isnan(t["slope"]) and np.isnan(slope)), (li, t["slope"], slope)
            assert (int(t["x"]), int(t["y"]), int(t["w"]), int(t["h"])) == box
            li += 1
        n_lines += len(lines)
        for k, i in enumerate(keep):
            g = res.group_bounds[i]
            assert (g["x"], g["y"], g["w"], g["h"], g["cx"], g["cy"]) == tuple(int(after[k][f]) for f in ("x", "y", "w", "h", "cx", "cy"))
    assert li == len(res.texts)
    assert list(res.group_all) == all_er
    return n_lines


@pytest.mark.gpu
@pytest.mark.parametrize("inner_sup", [False, True])
def test_gpu_group_stage_matches_oracle(S, cascade_paths, oracle, inner_sup):
    """text_detect through er_grouping (src/ER.cpp:33-69) on the reference's native 6 planes."""
    W, H, F = 640, 480, 3
    f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frames = np.stack([S.synth.stext_bgr(S.synth.frame_seed(40 + i), W, H) for i in range(F)])
    st = S.STAGE_ALL | S.STAGE_TRACK | S.STAGE_GROUP | (S.GROUP_INNER_SUP if inner_sup else 0)
    res = f.text_detect(frames, st)
    assert res.texts is not None and res.times[4] > 0
    groups = [np.nonzero(res.cands["frame"] == i)[0] for i in range(F)]
    n_lines = _check_lines(res, groups, oracle, inner_sup)
    assert n_lines > 0
    assert (res.texts["frame"] == np.sort(res.texts["frame"])).all()
    # single-stage entry point on image 0
    sel = groups[0]
    r1 = f.er_grouping(res.cands[sel], res.tracks[sel], inner_sup=inner_sup)
    n0 = int((res.texts["frame"] == 0).sum())
    assert len(r1.texts) == n0
    for k in range(n0):
        a, b = res.texts[k], r1.texts[k]
        assert list(res.text_ers[a["first"]:a["first"] + a["count"]] - sel[0]) == list(r1.text_ers[b["first"]:b["first"] + b["count"]])
        assert a["slope"] == b["slope"] or (np.isnan(a["slope"]) and np.isnan(b["slope"]))
    with pytest.raises(S.StrErError):
        f.text_detect(frames, S.STAGE_ALL | S.STAGE_GROUP)
    with pytest.raises(S.StrErError):
        f.text_detect(frames, S.STAGE_ALL | S.STAGE_TRACK | S.GROUP_OVERLAP_SUP)
    # er_grouping(tracked, text, overlap_sup = true, inner_sup) -- video_mode's call (src/utils.cpp:196) -- fused and single-stage
    res2 = f.text_detect(frames, st | S.GROUP_OVERLAP_SUP)
    assert res2.cands.tobytes() == res.cands.tobytes()
    _check_lines(res2, groups, oracle, inner_sup, overlap_sup=True)
    assert len(res2.group_all) <= len(res.group_all)
    r2 = f.er_grouping(res.cands[sel], res.tracks[sel], overlap_sup=True, inner_sup=inner_sup)
    r2.cands, r2.tracks = res.cands[sel], res.tracks[sel]
    _check_lines(r2, [np.arange(len(sel))], oracle, inner_sup, overlap_sup=True)
    f.close()


@pytest.mark.gpu
def test_gpu_group_random_boxes(erf, oracle, S):
    """er_grouping on made-up ERs: dense clusters with many equal centres, shared members, merges and NaN colours."""
    rng = np.random.default_rng(11)
    merged = 0
    for trial in range(6):
        n = int(rng.integers(1, 260))
        cd = np.zeros(n, S.CAND_DTYPE)
        tr = np.zeros(n, S.TRACK_DTYPE)
        cd["x"] = rng.integers(0, 300, n); cd["y"] = rng.integers(0, 60 if trial % 2 else 200, n)
        cd["w"] = rng.integers(4, 40, n); cd["h"] = rng.integers(6, 48, n)
        cd["area"] = (cd["w"].astype(np.int64) * cd["h"] * rng.uniform(0.3, 1.0, n)).astype(np.uint32) + 1
        cd["cls"] = rng.integers(0, 3, n)
        tr["cx"] = cd["x"] + cd["w"] // 2; tr["cy"] = cd["y"] + cd["h"] // 2
        for k in ("color1", "color2", "color3"):
            tr[k] = rng.integers(90, 130, n)
        tr["color2"][rng.random(n) < 0.03] = np.nan
        tr["tracked"] = (rng.random(n) < 0.8) & (cd["cls"] != 0)
        for inner, overlap in ((False, False), (True, False), (False, True), (True, True)):
            res = erf.er_grouping(cd, tr, overlap_sup=overlap, inner_sup=inner)
            res.cands, res.tracks = cd, tr
            _check_lines(res, [np.arange(n)], oracle, inner, overlap)
            if overlap and not inner:
                merged += int(tr["tracked"].sum()) - len(res.group_all)
    assert merged > 5                       # overlap_suppression did merge boxes away


@pytest.mark.gpu
def test_gpu_late_stages_on_empty_and_tiny_frames(S, cascade_paths):
    """No candidates at all (flat frames), frames smaller than a tile, one frame of pure noise: the late stages return empty
    tables, never garbage."""
    st = S.STAGE_ALL | S.STAGE_TRACK | S.STAGE_GROUP | S.GROUP_INNER_SUP
    for (w, h) in ((64, 48), (8, 8), (1, 1), (200, 3)):
        f = S.ERFilter(params=S.Params(max_width=w, max_height=h, max_frames=2))
        f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
        flat = np.full((2, h, w, 3), 128, np.uint8)
        res = f.text_detect(flat, st)
        assert len(res.tracks) == len(res.cands) and len(res.texts) == 0 and len(res.text_ers) == 0
        assert len(res.group_bounds) == len(res.cands) and len(res.group_all) == 0
        rng = np.random.default_rng(w * 1000 + h)
        noise = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        res = f.text_detect(noise, st)
        assert len(res.tracks) == len(res.cands)
        assert (res.tracks["tracked"][res.cands["cls"] == 0] == 0).all()
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,levels,mask,step", [(333, 201, 1, 0x3F, 8), (512, 300, 3, 0x07, 8), (257, 190, 2, 0x15, 4), (640, 360, 1, 0x38, 16)])
def test_gpu_track_and_group_on_odd_sizes(S, cascade_paths, oracle, W, H, levels, mask, step):
    """calc_color, er_track and er_grouping against the oracle on ragged frame sizes, channel subsets, pyramid levels, thresh steps."""
    F = 2
    f = S.ERFilter(params=S.Params(thresh_step=step, max_width=W, max_height=H, max_frames=F, n_pyr_levels=levels, channel_mask=mask))
    f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
    frames = np.stack([S.synth.stext_bgr(S.synth.frame_seed(70 + i), W, H) for i in range(F)])
    res = f.text_detect(frames, S.STAGE_ALL | S.STAGE_TRACK | S.STAGE_GROUP | S.GROUP_INNER_SUP)
    planes = {}
    for i in range(F):
        lvl = _ycrcb(oracle, frames[i])[:3]
        for l in range(levels):
            if l > 0:
                dw, dh = oracle.pyr_dims(W, H, l)
                lvl = np.stack([oracle.resize(p, dw, dh) for p in lvl])
            planes[(i, l)] = np.concatenate([lvl, 255 - lvl])
    groups = []
    for sel, order, e in _expected_tracks(oracle, res, planes):
        t = res.tracks[sel]
        live = e["cls"] != 0
        for k in ("color1", "color2", "color3"):
            assert np.array_equal(t[k][live], e[k][live], equal_nan=True)
        want = np.zeros(len(sel), bool)
        want[order] = True
        assert (t["tracked"].astype(bool) == want).all()
        groups.append(sel)
    _check_lines(res, groups, oracle, True)
    f.close()
