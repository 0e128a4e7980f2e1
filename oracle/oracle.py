"""ctypes front-end of the parity oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does (tests/test_layout.py enforces it).

`Oracle`     : oracle/liber_oracle.so, the plain-C restatement (er_oracle.c).
`RefCascade` : oracle/_ref/libref_adaboost.so, the reference's own
               src/adaboost.cpp compiled unmodified (built only where
               /root/reference exists; the prebuilt .so travels to the GPU box).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DBL_MAX = float(np.finfo(np.float64).max)


class _Node(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("level", "area", "x", "y", "w", "h", "parent", "child", "next", "key", "npix", "nsub")]


class _Tree(C.Structure):
    _fields_ = [("nodes", C.POINTER(_Node)), ("n_nodes", C.c_int32), ("root", C.c_int32),
                ("n_created", C.c_int64), ("dead_branch", C.c_int32)]


class _Params(C.Structure):
    _fields_ = [("thresh_step", C.c_int32), ("min_area", C.c_int32), ("max_area", C.c_int32),
                ("stability_t", C.c_int32), ("overlap_coef", C.c_double)]


NODE_DTYPE = np.dtype([(n, "<i4") for n in
                       ("level", "area", "x", "y", "w", "h", "parent", "child", "next", "key", "npix", "nsub")])


@dataclass
class Tree:
    nodes: np.ndarray          # structured array, NODE_DTYPE
    root: int
    n_created: int
    dead_branch: int


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(HERE, "liber_oracle.so")
    src = [os.path.join(HERE, f) for f in ("er_oracle.c", "er_oracle.h", "svm_oracle.c", "svm_oracle.h", "er_group_oracle.c",
                                           "er_group_oracle.h", "Makefile")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    ref_missing = os.path.exists("/root/reference/src/adaboost.cpp") and not all(
        os.path.exists(os.path.join(HERE, "_ref", f)) for f in ("libref_adaboost.so", "libref_svm.so", "svm-train"))
    if stale or ref_missing:
        subprocess.run(["make", "-s", "-C", HERE], check=True)


def _u8(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


class Oracle:
    def __init__(self) -> None:
        build()
        self.lib = L = C.CDLL(os.path.join(HERE, "liber_oracle.so"))
        u8p = C.POINTER(C.c_uint8)
        L.ero_quant_lut.argtypes = [C.c_int, u8p]
        L.ero_highest_level.argtypes = [C.c_int]
        L.ero_resize_linear_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        L.ero_compute_channels.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.ero_nv12_to_ycrcb.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.ero_tree_extract.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Tree)]
        L.ero_tree_bruteforce.argtypes = L.ero_tree_extract.argtypes
        L.ero_tree_free.argtypes = [C.POINTER(_Tree)]
        L.ero_nms.argtypes = [C.POINTER(_Tree), C.c_int, C.c_int, C.POINTER(_Params), C.c_int,
                              C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ero_aran26.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.ero_lbp24.argtypes = [u8p, u8p]
        L.ero_aran_dims.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ero_selftest_pow_vs_sqrt.argtypes = [C.c_int]
        L.ero_selftest_pow_vs_sqrt.restype = C.c_long
        L.ero_lbp_hist.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.ero_cascade_load.argtypes = [C.c_char_p]
        L.ero_cascade_load.restype = C.c_void_p
        L.ero_cascade_free.argtypes = [C.c_void_p]
        L.ero_cascade_n_stages.argtypes = [C.c_void_p]
        L.ero_cascade_n_stumps.argtypes = [C.c_void_p]
        L.ero_cascade_predict.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.ero_cascade_predict.restype = C.c_double
        L.ero_classify.argtypes = [u8p, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p,
                                   u8p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ero_pyr_dims.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ero_chain_features.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.ero_chain_features_slope.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_double, u8p]
        L.ero_ocr_normalise_slope.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_double, u8p]
        L.ero_rotate_mat.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ero_ocr_normalise.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.ero_chain_bitmaps.argtypes = [u8p, u8p]
        L.ero_otsu_threshold.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ero_calc_color.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.POINTER(C.c_double)]
        L.ero_er_track.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        ip, dp_ = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.ero_er_grouping.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip, ip, ip, ip, dp_, ip, C.c_int, ip, C.c_int]
        L.ero_fitline_avgslope.argtypes = [ip, ip, C.c_int]
        L.ero_fitline_avgslope.restype = C.c_double
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]

    # -- primitives -------------------------------------------------------
    def quant_lut(self, step: int) -> np.ndarray:
        lut = np.zeros(256, np.uint8)
        self.lib.ero_quant_lut(step, lut.ctypes.data_as(C.POINTER(C.c_uint8)))
        return lut

    def highest_level(self, step: int) -> int:
        return int(self.lib.ero_highest_level(step))

    def resize(self, src: np.ndarray, dw: int, dh: int) -> np.ndarray:
        src = _u8(src)
        dst = np.zeros((dh, dw), np.uint8)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_resize_linear_u8(src.ctypes.data_as(p), src.shape[1], src.shape[1], src.shape[0],
                                      dst.ctypes.data_as(p), dw, dw, dh)
        return dst

    def compute_channels(self, bgr: np.ndarray) -> np.ndarray:
        bgr = _u8(bgr)
        h, w, _ = bgr.shape
        out = np.zeros((6, h, w), np.uint8)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_compute_channels(bgr.ctypes.data_as(p), 3 * w, w, h, out.ctypes.data_as(p))
        return out

    def nv12_to_ycrcb(self, nv12: np.ndarray, w: int, h: int) -> np.ndarray:
        """(h * 3 // 2, w) NV12 frame -> (3, h, w) planes [Y, Cr, Cb] (build-defined ingest: chroma replicated 2x2)."""
        a = _u8(nv12).reshape(h + h // 2, w)
        out = np.zeros((3, h, w), np.uint8)
        p = C.POINTER(C.c_uint8)
        uv = a[h:]
        self.lib.ero_nv12_to_ycrcb(a.ctypes.data_as(p), w, uv.ctypes.data_as(p), w, w, h, out.ctypes.data_as(p))
        return out

    def pyr_dims(self, w0: int, h0: int, level: int):
        w, h = C.c_int(), C.c_int()
        self.lib.ero_pyr_dims(w0, h0, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def pyramid(self, plane: np.ndarray, n_levels: int):
        out = [_u8(plane)]
        h0, w0 = out[0].shape
        for k in range(1, n_levels):
            w, h = self.pyr_dims(w0, h0, k)
            out.append(self.resize(out[-1], w, h))
        return out

    # -- component tree ---------------------------------------------------
    def _tree(self, fn, img: np.ndarray, step: int, min_area: int) -> Tree:
        img = _u8(img)
        h, w = img.shape
        t = _Tree()
        rc = fn(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, w, h, step, min_area, C.byref(t))
        if rc != 0:
            raise RuntimeError("oracle tree extraction failed")
        buf = (C.c_char * (C.sizeof(_Node) * t.n_nodes)).from_address(C.addressof(t.nodes.contents))
        nodes = np.frombuffer(buf, dtype=NODE_DTYPE).copy()
        out = Tree(nodes, int(t.root), int(t.n_created), int(t.dead_branch))
        self.lib.ero_tree_free(C.byref(t))
        return out

    def tree_extract(self, img, step=8, min_area=120) -> Tree:
        return self._tree(self.lib.ero_tree_extract, img, step, min_area)

    def tree_bruteforce(self, img, step=8, min_area=120) -> Tree:
        return self._tree(self.lib.ero_tree_bruteforce, img, step, min_area)

    def nms(self, tree: Tree, rows: int, cols: int, step=8, min_area=120, max_area=900000,
            stability_t=2, overlap_coef=0.7, sibling_mode=0):
        nodes = np.ascontiguousarray(tree.nodes)
        t = _Tree(nodes.ctypes.data_as(C.POINTER(_Node)), len(nodes), tree.root, 0, 0)
        prm = _Params(step, min_area, max_area, stability_t, overlap_coef)
        pool = C.POINTER(C.c_int32)()
        n, amb = C.c_int32(), C.c_int32()
        rc = self.lib.ero_nms(C.byref(t), rows, cols, C.byref(prm), sibling_mode,
                              C.byref(pool), C.byref(n), C.byref(amb))
        if rc != 0:
            raise RuntimeError("oracle nms failed")
        out = np.array([pool[i] for i in range(n.value)], dtype=np.int32)
        self._libc.free(C.cast(pool, C.c_void_p))
        return out, int(amb.value)

    # -- classify ---------------------------------------------------------
    def aran26(self, roi: np.ndarray) -> np.ndarray:
        roi = _u8(roi)
        tile = np.zeros((26, 26), np.uint8)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_aran26(roi.ctypes.data_as(p), roi.shape[1], roi.shape[1], roi.shape[0], tile.ctypes.data_as(p))
        return tile

    def aran_dims(self, w: int, h: int):
        dw, dh = C.c_int(), C.c_int()
        self.lib.ero_aran_dims(w, h, C.byref(dw), C.byref(dh))
        return dw.value, dh.value

    def lbp24(self, tile: np.ndarray) -> np.ndarray:
        tile = _u8(tile)
        out = np.zeros((24, 24), np.uint8)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_lbp24(tile.ctypes.data_as(p), out.ctypes.data_as(p))
        return out

    def lbp_hist(self, roi: np.ndarray) -> np.ndarray:
        roi = _u8(roi)
        out = np.zeros(1024, np.float64)
        self.lib.ero_lbp_hist(roi.ctypes.data_as(C.POINTER(C.c_uint8)), roi.shape[1], roi.shape[1], roi.shape[0],
                              out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def chain_features(self, roi: np.ndarray, slope: float = 0.0) -> np.ndarray:
        roi = _u8(roi)
        q = np.zeros(1800, np.uint8)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_chain_features_slope(roi.ctypes.data_as(p), roi.shape[1], roi.shape[1], roi.shape[0], float(slope), q.ctypes.data_as(p))
        return q

    def rotate_mat(self, img: np.ndarray, rad: float, crop: bool = True) -> np.ndarray:
        """OCR::rotate_mat (src/OCR.cpp:254-357)."""
        img = _u8(img)
        p = C.POINTER(C.c_uint8)
        dst = p()
        dw, dh = C.c_int(0), C.c_int(0)
        self.lib.ero_rotate_mat(img.ctypes.data_as(p), img.shape[1], img.shape[0], float(rad), int(crop), C.byref(dst), C.byref(dw), C.byref(dh))
        out = np.ctypeslib.as_array(dst, shape=(dh.value, dw.value)).copy()
        self._libc.free(C.cast(dst, C.c_void_p))
        return out

    def ocr_normalise(self, roi: np.ndarray, slope: float = 0.0) -> np.ndarray:
        roi = _u8(roi)
        out = np.zeros((30, 30), np.uint8)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_ocr_normalise_slope(roi.ctypes.data_as(p), roi.shape[1], roi.shape[1], roi.shape[0], float(slope), out.ctypes.data_as(p))
        return out

    # -- er_group_oracle.c ---------------------------------
    ER_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("cx", "<i4"), ("cy", "<i4"), ("area", "<i4"),
                         ("ch", "<i4"), ("cls", "<i4"), ("id", "<i4"), ("color1", "<f8"), ("color2", "<f8"), ("color3", "<f8")])

    def calc_color(self, mask_plane: np.ndarray, color_img: np.ndarray, box) -> np.ndarray:
        m, ci = _u8(mask_plane), _u8(color_img)
        out = np.zeros(3, np.float64)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_calc_color(m.ctypes.data_as(p), m.shape[1], int(box[0]), int(box[1]), int(box[2]), int(box[3]),
                                ci.ctypes.data_as(p), ci.shape[1] * 3, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def er_track(self, ers: np.ndarray):
        """ers: ER_DTYPE array (x,y,w,h,area,cls,color1-3 filled).  Returns (order of all_er, ers with cx/cy set)."""
        e = np.ascontiguousarray(ers, dtype=self.ER_DTYPE).copy()
        order = np.zeros(len(e) + 1, np.int32)
        n = self.lib.ero_er_track(e.ctypes.data_as(C.c_void_p), len(e), order.ctypes.data_as(C.POINTER(C.c_int)))
        return order[:n].copy(), e

    def er_grouping(self, ers: np.ndarray, overlap_sup: bool = False, inner_sup: bool = False):
        """ers: all_er as an ER_DTYPE array (cx, cy, colours set).  Returns (all_idx after sort/suppression,
        lines = [(member indices into ers, slope, box xywh)], ers as the call leaves them)."""
        e = np.ascontiguousarray(ers, dtype=self.ER_DTYPE).copy()
        n = len(e)
        cap_l, cap_m = n + 1, n * n + 2 * n + 2
        all_idx = np.zeros(n + 1, np.int32)
        n_all = C.c_int(0)
        first, count = np.zeros(cap_l, np.int32), np.zeros(cap_l, np.int32)
        slope, box, member = np.zeros(cap_l, np.float64), np.zeros(4 * cap_l, np.int32), np.zeros(cap_m, np.int32)
        ip, dp_ = C.POINTER(C.c_int), C.POINTER(C.c_double)
        nl = self.lib.ero_er_grouping(e.ctypes.data_as(C.c_void_p), n, int(overlap_sup), int(inner_sup), all_idx.ctypes.data_as(ip),
                                      C.byref(n_all), first.ctypes.data_as(ip), count.ctypes.data_as(ip), slope.ctypes.data_as(dp_),
                                      box.ctypes.data_as(ip), cap_l, member.ctypes.data_as(ip), cap_m)
        assert nl >= 0, "oracle er_grouping overflow"
        lines = [(member[first[k]:first[k] + count[k]].copy(), float(slope[k]), tuple(int(v) for v in box[4 * k:4 * k + 4])) for k in range(nl)]
        return all_idx[:n_all.value].copy(), lines, e

    def fitline_avgslope(self, pts) -> float:
        px = np.ascontiguousarray([p[0] for p in pts], dtype=np.int32)
        py = np.ascontiguousarray([p[1] for p in pts], dtype=np.int32)
        ip = C.POINTER(C.c_int)
        return float(self.lib.ero_fitline_avgslope(px.ctypes.data_as(ip), py.ctypes.data_as(ip), len(px)))

    def otsu(self, img: np.ndarray, invert: bool = False) -> int:
        img = _u8(img)
        return int(self.lib.ero_otsu_threshold(img.ctypes.data_as(C.POINTER(C.c_uint8)), img.shape[1], img.shape[1], img.shape[0], int(invert)))

    def chain_bitmaps(self, img30: np.ndarray) -> np.ndarray:
        img30 = _u8(img30)
        m = np.zeros((8, 30, 30), np.uint8)
        p = C.POINTER(C.c_uint8)
        self.lib.ero_chain_bitmaps(img30.ctypes.data_as(p), m.ctypes.data_as(p))
        return m

    def cascade_load(self, path: str) -> "OracleCascade":
        h = self.lib.ero_cascade_load(path.encode())
        if not h:
            raise FileNotFoundError(path)
        return OracleCascade(self, h)

    def classify(self, plane: np.ndarray, boxes: np.ndarray, strong: "OracleCascade", weak: "OracleCascade"):
        plane = _u8(plane)
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        n = len(boxes)
        cls = np.zeros(n, np.uint8)
        ss = np.zeros(n, np.float64)
        sw = np.zeros(n, np.float64)
        self.lib.ero_classify(plane.ctypes.data_as(C.POINTER(C.c_uint8)), plane.shape[1],
                              boxes.ctypes.data_as(C.POINTER(C.c_int32)), n, strong.h, weak.h,
                              cls.ctypes.data_as(C.POINTER(C.c_uint8)),
                              ss.ctypes.data_as(C.POINTER(C.c_double)), sw.ctypes.data_as(C.POINTER(C.c_double)))
        return cls, ss, sw

    # -- whole per-plane hot path (src/ER.cpp:50-60 loop body) -------------
    def detect_plane(self, plane: np.ndarray, strong=None, weak=None, step=8, min_area=120, max_area=900000,
                     stability_t=2, overlap_coef=0.7, sibling_mode=0):
        plane = _u8(plane)
        rows, cols = plane.shape
        tree = self.tree_extract(plane, step, min_area)
        pool, amb = self.nms(tree, rows, cols, step, min_area, max_area, stability_t, overlap_coef, sibling_mode)
        res = {"tree": tree, "pool": pool, "ambiguous": amb}
        if strong is not None and weak is not None:
            nd = tree.nodes[pool]
            boxes = np.stack([nd["x"], nd["y"], nd["w"], nd["h"]], axis=1).astype(np.int32) if len(pool) else \
                np.zeros((0, 4), np.int32)
            res["cls"], res["s_strong"], res["s_weak"] = self.classify(plane, boxes, strong, weak)
        return res


class OracleCascade:
    def __init__(self, o: Oracle, h) -> None:
        self.o, self.h = o, h

    @property
    def n_stages(self) -> int:
        return int(self.o.lib.ero_cascade_n_stages(self.h))

    @property
    def n_stumps(self) -> int:
        return int(self.o.lib.ero_cascade_n_stumps(self.h))

    def predict(self, fv: np.ndarray) -> float:
        fv = np.ascontiguousarray(fv, dtype=np.float64)
        assert fv.size == 1024
        return float(self.o.lib.ero_cascade_predict(self.h, fv.ctypes.data_as(C.POINTER(C.c_double))))

    def __del__(self):
        try:
            self.o.lib.ero_cascade_free(self.h)
        except Exception:
            pass


class RefCascade:
    """The reference's own CascadeBoost (oracle/_ref).  `available()` is False
    when the prebuilt library is absent and /root/reference is not mounted."""

    _lib = None

    @classmethod
    def available(cls) -> bool:
        build()
        return os.path.exists(os.path.join(HERE, "_ref", "libref_adaboost.so"))

    def __init__(self, path: str) -> None:
        if RefCascade._lib is None:
            L = C.CDLL(os.path.join(HERE, "_ref", "libref_adaboost.so"))
            L.ref_cascade_load.argtypes = [C.c_char_p]
            L.ref_cascade_load.restype = C.c_void_p
            L.ref_cascade_predict.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
            L.ref_cascade_predict.restype = C.c_double
            L.ref_cascade_n_stumps.argtypes = [C.c_void_p]
            L.ref_cascade_free.argtypes = [C.c_void_p]
            RefCascade._lib = L
        self.h = RefCascade._lib.ref_cascade_load(path.encode())
        if not self.h:
            raise FileNotFoundError(path)

    @property
    def n_stumps(self) -> int:
        return int(RefCascade._lib.ref_cascade_n_stumps(self.h))

    def predict(self, fv: np.ndarray) -> float:
        fv = np.ascontiguousarray(fv, dtype=np.float64)
        return float(RefCascade._lib.ref_cascade_predict(self.h, fv.ctypes.data_as(C.POINTER(C.c_double)), fv.size))


class OracleSVM:
    """oracle/svm_oracle.c: plain-C restatement of the reference's libsvm inference."""

    def __init__(self, o: Oracle, path: str) -> None:
        L = o.lib
        L.ero_svm_load.restype = C.c_void_p
        L.ero_svm_load.argtypes = [C.c_char_p]
        L.ero_svm_free.argtypes = [C.c_void_p]
        for f in (L.ero_svm_nr_class, L.ero_svm_total_sv, L.ero_svm_max_index):
            f.argtypes = [C.c_void_p]
        L.ero_svm_predict_probability.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_double)]
        self.L = L
        self.h = L.ero_svm_load(path.encode())
        if not self.h:
            raise FileNotFoundError(path)
        self.k, self.l = int(L.ero_svm_nr_class(self.h)), int(L.ero_svm_total_sv(self.h))

    def predict_probability(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float64)
        dec = np.zeros(self.k * (self.k - 1) // 2)
        prob = np.zeros(self.k)
        p = C.POINTER(C.c_double)
        lab = self.L.ero_svm_predict_probability(self.h, x.ctypes.data_as(p), x.size, dec.ctypes.data_as(p), prob.ctypes.data_as(p))
        return int(lab), prob, dec


class _SvmNode(C.Structure):
    _fields_ = [("index", C.c_int), ("value", C.c_double)]


class RefSVM:
    """The reference's vendored libsvm 3.21 itself (oracle/_ref/libref_svm.so, its own extern "C" API)."""

    @staticmethod
    def available() -> bool:
        build()
        return os.path.exists(os.path.join(HERE, "_ref", "libref_svm.so"))

    def __init__(self, path: str) -> None:
        L = C.CDLL(os.path.join(HERE, "_ref", "libref_svm.so"))
        L.svm_load_model.restype = C.c_void_p
        L.svm_load_model.argtypes = [C.c_char_p]
        L.svm_get_nr_class.argtypes = [C.c_void_p]
        L.svm_predict_probability.restype = C.c_double
        L.svm_predict_probability.argtypes = [C.c_void_p, C.POINTER(_SvmNode), C.POINTER(C.c_double)]
        L.svm_predict_values.restype = C.c_double
        L.svm_predict_values.argtypes = [C.c_void_p, C.POINTER(_SvmNode), C.POINTER(C.c_double)]
        self.L = L
        self.h = L.svm_load_model(path.encode())
        if not self.h:
            raise FileNotFoundError(path)
        self.k = int(L.svm_get_nr_class(self.h))

    def predict_probability(self, x: np.ndarray):
        nz = np.nonzero(x)[0]
        nodes = (_SvmNode * (len(nz) + 1))()
        for j, i in enumerate(nz):
            nodes[j].index, nodes[j].value = int(i), float(x[i])
        nodes[len(nz)].index = -1
        pv = (C.c_double * self.k)()
        dv = (C.c_double * (self.k * (self.k - 1) // 2))()
        lab = int(self.L.svm_predict_probability(self.h, nodes, pv))
        self.L.svm_predict_values(self.h, nodes, dv)
        return lab, np.array(list(pv)), np.array(list(dv))
