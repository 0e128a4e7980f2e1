"""ctypes binding of include/str_er.h.

`ERFilter` mirrors the reference's `class ERFilter` (inc/ER.h:110-169) for the hot path:
same constructor arguments and method names (`text_detect`, `compute_channels`,
`er_tree_extract`, `non_maximum_supression`, `classify`, `make_LBP_hist`,
`set_thresh_step`, `set_min_area`), results as numpy structured arrays instead of
heap `ER*` trees.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

STAGE_EXTRACT, STAGE_NMS, STAGE_CLASSIFY, STAGE_ALL, STAGE_OCR, WANT_NODES, STAGE_TRACK = 1, 2, 4, 7, 8, 16, 32
STAGE_GROUP, GROUP_INNER_SUP, STAGE_OCR_LINES, GROUP_OVERLAP_SUP = 64, 128, 256, 512
TEXT_DTYPE = np.dtype([("frame", "<u4"), ("pyr", "u1"), ("r0", "u1"), ("r1", "u1"), ("r2", "u1"), ("first", "<i4"), ("count", "<i4"),
                       ("slope", "<f8"), ("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4")])
GBOUND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("cx", "<i4"), ("cy", "<i4")])
TRACK_DTYPE = np.dtype([("color1", "<f8"), ("color2", "<f8"), ("color3", "<f8"), ("cx", "<i4"), ("cy", "<i4"),
                        ("tracked", "<u4"), ("reserved", "<u4")])
CLS_POOL, CLS_STRONG, CLS_WEAK = 0, 1, 2
MEM_HOST, MEM_DEVICE = 0, 1

NODE_DTYPE = np.dtype([("key", "<u4"), ("parent", "<i4"), ("area", "<i4"), ("x", "<u2"), ("y", "<u2"),
                       ("w", "<u2"), ("h", "<u2"), ("level", "u1"), ("flags", "u1"), ("reserved", "<u2")])
CAND_DTYPE = np.dtype([("frame", "<u4"), ("ch", "u1"), ("pyr", "u1"), ("level", "u1"), ("cls", "u1"),
                       ("x", "<u2"), ("y", "<u2"), ("w", "<u2"), ("h", "<u2"), ("area", "<u4"), ("key", "<u4"),
                       ("node", "<i4"), ("plane", "<u4"), ("score_strong", "<f8"), ("score_weak", "<f8")])
PLANE_DTYPE = np.dtype([("frame", "<u4"), ("ch", "u1"), ("pyr", "u1"), ("r0", "u1"), ("r1", "u1"), ("width", "<i4"),
                        ("height", "<i4"), ("n_created", "<i4"), ("n_kept", "<i4"), ("n_pool", "<i4"), ("n_strong", "<i4"),
                        ("n_weak", "<i4"), ("ambiguous", "<i4"), ("root", "<i4")])
assert NODE_DTYPE.itemsize == 24 and CAND_DTYPE.itemsize == 48 and PLANE_DTYPE.itemsize == 44


class StrErError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"str_er error {code}: {msg}")
        self.code = code


class _Params(C.Structure):
    _fields_ = [("thresh_step", C.c_int32), ("min_area", C.c_int32), ("max_area", C.c_int32),
                ("stability_t", C.c_int32), ("overlap_coef", C.c_double), ("n_pyr_levels", C.c_int32),
                ("channel_mask", C.c_uint32), ("device", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_frames", C.c_int32), ("kept_cap", C.c_int32),
                ("pool_cap", C.c_int32), ("sibling_order", C.c_int32), ("stream", C.c_void_p)]


class _PlaneInfo(C.Structure):
    _fields_ = [("frame", C.c_uint32), ("ch", C.c_uint8), ("pyr", C.c_uint8), ("r0", C.c_uint8), ("r1", C.c_uint8),
                ("width", C.c_int32), ("height", C.c_int32), ("n_created", C.c_int32), ("n_kept", C.c_int32),
                ("n_pool", C.c_int32), ("n_strong", C.c_int32), ("n_weak", C.c_int32), ("ambiguous", C.c_int32),
                ("root", C.c_int32)]


@dataclass
class Params:
    """Constructor arguments of ERFilter (inc/ER.h:113; src/main.cpp:22) + capacity."""
    thresh_step: int = 8
    min_area: int = 120
    max_area: int = 900000
    stability_t: int = 2
    overlap_coef: float = 0.7
    n_pyr_levels: int = 1
    channel_mask: int = 0x3F
    device: int = 0
    max_width: int = 1920
    max_height: int = 1080
    max_frames: int = 8
    kept_cap: int = 0
    pool_cap: int = 0
    sibling_order: int = 0
    stream: Optional[int] = None


_LIB = None


def lib_path() -> str:
    # STR_ER_LIB: developer switch, load another build of the same library (tools/dev_stop_all.sh)
    return os.environ.get("STR_ER_LIB") or os.path.join(HERE, "lib", "libstr_er_hip.so")


def load_library():
    """Load libstr_er_hip.so; raises if it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: run `python scene-text-recognition_amd/build.py` (hipcc, gfx950). "
            "There is no CPU implementation of this path.")
    L = C.CDLL(path)
    u8p, i32p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    vp = C.c_void_p
    L.str_er_abi_version.restype = C.c_int
    L.str_er_default_params.argtypes = [C.POINTER(_Params)]
    L.str_er_create.argtypes = [C.POINTER(_Params), C.POINTER(vp)]
    L.str_er_destroy.argtypes = [vp]
    L.str_er_last_error.argtypes = [vp]
    L.str_er_last_error.restype = C.c_char_p
    L.str_er_strerror.argtypes = [C.c_int]
    L.str_er_strerror.restype = C.c_char_p
    L.str_er_set_thresh_step.argtypes = [vp, C.c_int32]
    L.str_er_set_min_area.argtypes = [vp, C.c_int32]
    L.str_er_load_cascade.argtypes = [vp, C.c_int, C.c_char_p]
    L.str_er_load_cascade_mem.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    L.str_er_cascade_info.argtypes = [vp, C.c_int, i32p, i32p]
    L.str_er_detect_bgr.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int,
                                    C.c_uint32, C.POINTER(vp)]
    L.str_er_detect_bgr_planes.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int,
                                           C.c_uint32, vp, C.c_int32, C.POINTER(vp)]
    L.str_er_strip_extract.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.str_er_strip_free.argtypes = [vp]
    L.str_er_strip_free.restype = None
    L.str_er_strip_merge.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int, vp, vp, C.c_int32, C.c_uint32, C.POINTER(vp)]
    L.str_er_strip_extract_dev.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.str_er_strip_merge_ex.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int32, vp, C.c_uint32, C.POINTER(vp)]
    L.str_er_comm_allgather_bytes.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.str_er_comm_free.argtypes = [vp]
    L.str_er_comm_free.restype = None
    L.str_er_detect_planes.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int,
                                       C.c_uint32, C.POINTER(vp)]
    L.str_er_compute_channels.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp]
    L.str_er_classify_boxes.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, vp, vp, vp]
    L.str_er_lbp_hist.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, vp, vp]
    L.str_er_calc_lbp.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, vp]
    L.str_er_cascade_predict.argtypes = [vp, C.c_int, vp, C.c_int32, vp]
    L.str_er_load_svm_model.argtypes = [vp, C.c_char_p, C.c_int32]
    L.str_er_load_svm_model_mem.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_int32]
    L.str_er_svm_info.argtypes = [vp, i32p, i32p, i32p]
    L.str_er_svm_forms.argtypes = [vp, i32p, i32p]
    L.str_er_svm_predict_probability.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.str_er_ocr_chain_run.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, vp, vp, vp]
    L.str_er_ocr_chain_run_slope.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, vp, C.c_int32, vp, vp, vp]
    L.str_er_nms_tree.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, i32p, i32p]
    L.str_er_nms_tree_plane.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, i32p, i32p]
    L.str_er_flood_order.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp]
    L.str_er_comm_unique_id.argtypes = [vp]
    L.str_er_comm_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, C.POINTER(vp)]
    L.str_er_comm_local_group.argtypes = [C.c_int32, C.POINTER(vp)]
    L.str_er_comm_create_local.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.str_er_comm_local_group_free.argtypes = [vp]
    L.str_er_comm_local_group_free.restype = None
    L.str_er_comm_destroy.argtypes = [vp]
    L.str_er_comm_destroy.restype = None
    L.str_er_comm_last_error.argtypes = [vp]
    L.str_er_comm_last_error.restype = C.c_char_p
    L.str_er_gather_cands.argtypes = [vp, vp, C.c_int32, C.c_uint32, C.POINTER(vp), i32p, i32p]
    L.str_er_gather_last.argtypes = [vp, vp, C.c_uint32, C.POINTER(vp), i32p, i32p]
    L.str_er_gather_free.argtypes = [vp]
    L.str_er_gather_free.restype = None
    L.str_er_resize_plane.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, C.c_int32]
    L.str_er_result_n_planes.argtypes = [vp]
    L.str_er_result_n_planes.restype = C.c_int32
    L.str_er_result_plane_info.argtypes = [vp, C.c_int32, C.POINTER(_PlaneInfo)]
    L.str_er_result_plane_infos.argtypes = [vp, i32p]
    L.str_er_result_plane_infos.restype = vp
    L.str_er_result_cands.argtypes = [vp, i32p]
    L.str_er_result_cands.restype = vp
    L.str_er_result_plane_cands.argtypes = [vp, C.c_int32, i32p]
    L.str_er_result_plane_cands.restype = vp
    L.str_er_result_plane_nodes.argtypes = [vp, C.c_int32, i32p]
    L.str_er_result_plane_nodes.restype = vp
    L.str_er_result_tracks.argtypes = [vp, i32p]
    L.str_er_result_tracks.restype = vp
    L.str_er_set_min_ocr_prob.argtypes = [vp, C.c_double]
    for fn in (L.str_er_result_texts, L.str_er_result_text_ers, L.str_er_result_group_bounds, L.str_er_result_group_all,
               L.str_er_result_line_labels, L.str_er_result_line_probs, L.str_er_result_line_kept, L.str_er_result_text_alive):
        fn.argtypes = [vp, i32p]
        fn.restype = vp
    L.str_er_er_grouping.argtypes = [vp, vp, vp, C.c_int32, C.c_int, C.c_int, C.POINTER(vp)]
    L.str_er_stream_create.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.str_er_stream_destroy.argtypes = [vp]
    L.str_er_stream_destroy.restype = None
    L.str_er_stream_depth.argtypes = [vp]
    L.str_er_stream_context.argtypes = [vp, C.c_int32]
    L.str_er_stream_context.restype = vp
    L.str_er_stream_last_error.argtypes = [vp]
    L.str_er_stream_last_error.restype = C.c_char_p
    L.str_er_stream_load_cascade.argtypes = [vp, C.c_int, C.c_char_p]
    L.str_er_stream_acquire.argtypes = [vp, i32p, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.str_er_stream_submit.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.str_er_stream_submit_nv12.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.str_er_detect_nv12.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.str_er_stream_submit_copy.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.str_er_stream_next.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.str_er_stream_pending.argtypes = [vp]
    L.str_er_calc_color.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, vp]
    L.str_er_er_track.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp]
    L.str_er_result_ocr_labels.argtypes = [vp, i32p]
    L.str_er_result_ocr_labels.restype = vp
    L.str_er_result_ocr_probs.argtypes = [vp, i32p]
    L.str_er_result_ocr_probs.restype = vp
    L.str_er_result_times.argtypes = [vp]
    L.str_er_result_times.restype = f64p
    L.str_er_result_cands_to_device.argtypes = [vp, vp, vp, C.c_int32, i32p]
    L.str_er_result_free.argtypes = [vp]
    L.str_er_last_tree_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.str_er_tile2_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.str_er_ocr_stage_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.str_er_last_profile.argtypes = [vp, C.POINTER(C.c_char_p), f64p, C.c_int32]
    L.str_er_set_profiling.argtypes = [vp, C.c_int]
    L.str_er_workspace_bytes.argtypes = [vp]
    L.str_er_workspace_bytes.restype = C.c_int64
    L.str_er_runtime_hint.restype = C.c_char_p
    L.str_er_tie_stats.argtypes = [vp, C.POINTER(C.c_uint64), f64p, i32p]
    if L.str_er_abi_version() != 2:
        raise RuntimeError("libstr_er_hip.so ABI version mismatch")
    _LIB = L
    return L


def set_batch_slots(n: int) -> int:
    """At most n detect calls of the process have their batch's kernels on the GPU at a time (0: no limit); include/str_er.h."""
    return int(load_library().str_er_set_batch_slots(int(n)))


def apply_runtime_hint() -> int:
    """Opt in to the HIP runtime settings the library recommends (str_er_runtime_hint(): more hardware queues).  Only
    effective before the process's first HIP call (e.g. before torch.cuda is initialised); importing the package does
    not touch the environment."""
    return int(load_library().str_er_apply_runtime_hint())


@dataclass
class PlaneResult:
    frame: int
    ch: int
    pyr: int
    width: int
    height: int
    n_created: int
    n_kept: int
    n_pool: int
    n_strong: int
    n_weak: int
    ambiguous: int
    root: int
    cands: np.ndarray                      # CAND_DTYPE, ascending key
    nodes: Optional[np.ndarray] = None     # NODE_DTYPE, ascending (key, level)

    @property
    def pool(self) -> np.ndarray:
        return self.cands

    @property
    def strong(self) -> np.ndarray:
        return self.cands[self.cands["cls"] == CLS_STRONG]

    @property
    def weak(self) -> np.ndarray:
        return self.cands[self.cands["cls"] == CLS_WEAK]


class Result:
    """Outcome of one detect call.  `info` (PLANE_DTYPE) and `cands` (CAND_DTYPE) are flat arrays;
    `planes` builds one PlaneResult per plane on first use."""

    def __init__(self, info: np.ndarray, cands: np.ndarray, times: np.ndarray, profile: dict, nodes=None):
        self.info, self.cands, self.times, self.profile, self._nodes = info, cands, times, profile, nodes
        self.ocr_label = None      # with STAGE_OCR: per candidate, -1 for cls == 0
        self.ocr_prob = None
        self.tracks = None         # with STAGE_TRACK: TRACK_DTYPE per candidate (zeros for cls == 0)
        self.texts = None          # with STAGE_GROUP: TEXT_DTYPE per line; members = text_ers[first:first+count] (candidate indices)
        self.text_ers = None
        self.group_bounds = None   # GBOUND_DTYPE per candidate: bound / center as er_grouping leaves them
        self.group_all = None      # all_er after er_grouping's sort / inner_suppression (candidate indices, images concatenated)
        self.line_label = None     # with STAGE_OCR_LINES: per entry of text_ers, chain_run's label / prob with the line's slope,
        self.line_prob = None      # whether the member survives er_ocr's two deletions, and per line whether >= 2 members do
        self.line_kept = None
        self.text_alive = None
        self._planes = None

    @property
    def planes(self) -> List[PlaneResult]:
        if self._planes is None:
            out, off = [], 0
            for i, pi in enumerate(self.info):
                n = int(pi["n_pool"])
                out.append(PlaneResult(int(pi["frame"]), int(pi["ch"]), int(pi["pyr"]), int(pi["width"]), int(pi["height"]),
                                       int(pi["n_created"]), int(pi["n_kept"]), n, int(pi["n_strong"]), int(pi["n_weak"]),
                                       int(pi["ambiguous"]), int(pi["root"]), self.cands[off:off + n],
                                       self._nodes[i] if self._nodes is not None else None))
                off += n
            self._planes = out
        return self._planes


def _np_ptr(a: np.ndarray) -> int:
    return a.ctypes.data


class ERFilter:
    """Drop-in for the hot-path surface of the reference's ERFilter (inc/ER.h:110-136)."""

    def __init__(self, thresh_step: int = 2, min_area: int = 100, max_area: int = 100000, stability_t: int = 2,
                 overlap_coef: float = 0.7, min_ocr_prob: float = 0.01, *, params: Optional[Params] = None, **cap):
        # positional defaults are the reference's (inc/ER.h:113); src/main.cpp:22 passes 8,120,900000,2,0.7,0.15
        self.L = load_library()
        p = params or Params(thresh_step=thresh_step, min_area=min_area, max_area=max_area, stability_t=stability_t,
                             overlap_coef=overlap_coef, **cap)
        self.params = p
        self.min_ocr_prob = min_ocr_prob
        cp = _Params(p.thresh_step, p.min_area, p.max_area, p.stability_t, p.overlap_coef, p.n_pyr_levels,
                     p.channel_mask, p.device, p.max_width, p.max_height, p.max_frames, p.kept_cap, p.pool_cap,
                     p.sibling_order, p.stream)
        h = C.c_void_p()
        rc = self.L.str_er_create(C.byref(cp), C.byref(h))
        if rc != 0:
            raise StrErError(rc, (self.L.str_er_last_error(None) or b"").decode())
        self.h = h
        self.stc = None  # names of the reference's public members (inc/ER.h:117-118)
        self.wtc = None

    # ---- lifetime -------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "h", None):
            self.L.str_er_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise StrErError(rc, (self.L.str_er_last_error(self.h) or b"").decode())

    # ---- models: stc / wtc = new CascadeBoost(file)  (src/main.cpp:23-24) ------------------
    def load_cascade(self, which: int, path: str) -> None:
        self._check(self.L.str_er_load_cascade(self.h, which, path.encode()))
        if which == 0:
            self.stc = path
        else:
            self.wtc = path

    def load_cascade_text(self, which: int, text: str) -> None:
        b = text.encode()
        self._check(self.L.str_er_load_cascade_mem(self.h, which, b, len(b)))

    def cascade_info(self, which: int):
        a, b = C.c_int32(), C.c_int32()
        self._check(self.L.str_er_cascade_info(self.h, which, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_thresh_step(self, t: int) -> None:      # src/ER.cpp:21-24
        self._check(self.L.str_er_set_thresh_step(self.h, t))
        self.params.thresh_step = t

    def set_min_area(self, m: int) -> None:         # src/ER.cpp:27-30
        self._check(self.L.str_er_set_min_area(self.h, m))
        self.params.min_area = m

    # ---- results ------------------------------------------------------------------------------
    def _collect(self, rh: C.c_void_p, profile: Optional[dict] = None) -> Result:
        L = self.L
        try:
            n = C.c_int32()
            ptr = L.str_er_result_cands(rh, C.byref(n))
            if n.value:
                cands = np.frombuffer((C.c_char * (48 * n.value)).from_address(ptr), dtype=CAND_DTYPE).copy()
            else:
                cands = np.zeros(0, CAND_DTYPE)
            ptr = L.str_er_result_plane_infos(rh, C.byref(n))
            info = np.frombuffer((C.c_char * (44 * n.value)).from_address(ptr), dtype=PLANE_DTYPE).copy()
            nodes = None
            nn = C.c_int32()
            if n.value and L.str_er_result_plane_nodes(rh, 0, C.byref(nn)):
                nodes = []
                for i in range(n.value):
                    nptr = L.str_er_result_plane_nodes(rh, i, C.byref(nn))
                    nodes.append(np.frombuffer((C.c_char * (24 * nn.value)).from_address(nptr), dtype=NODE_DTYPE).copy()
                                 if nn.value else np.zeros(0, NODE_DTYPE))
            t = L.str_er_result_times(rh)
            times = np.array([t[i] for i in range(7)])
            res = Result(info, cands, times, self.last_profile() if profile is None else profile, nodes)
            no = C.c_int32()
            lp = L.str_er_result_ocr_labels(rh, C.byref(no))
            if lp:
                res.ocr_label = np.frombuffer((C.c_char * (4 * no.value)).from_address(lp), dtype=np.int32).copy()
                pp = L.str_er_result_ocr_probs(rh, C.byref(no))
                res.ocr_prob = np.frombuffer((C.c_char * (8 * no.value)).from_address(pp), dtype=np.float64).copy()
            tp = L.str_er_result_tracks(rh, C.byref(no))
            if tp:
                res.tracks = (np.frombuffer((C.c_char * (40 * no.value)).from_address(tp), dtype=TRACK_DTYPE).copy()
                              if no.value else np.zeros(0, TRACK_DTYPE))
            xp = L.str_er_result_texts(rh, C.byref(no))
            if xp:
                res.texts = (np.frombuffer((C.c_char * (40 * no.value)).from_address(xp), dtype=TEXT_DTYPE).copy()
                             if no.value else np.zeros(0, TEXT_DTYPE))
                ep = L.str_er_result_text_ers(rh, C.byref(no))
                res.text_ers = (np.frombuffer((C.c_char * (4 * no.value)).from_address(ep), dtype=np.int32).copy()
                                if no.value else np.zeros(0, np.int32))
                ap = L.str_er_result_group_all(rh, C.byref(no))
                res.group_all = (np.frombuffer((C.c_char * (4 * no.value)).from_address(ap), dtype=np.int32).copy()
                                 if no.value else np.zeros(0, np.int32))
                bp = L.str_er_result_group_bounds(rh, C.byref(no))
                res.group_bounds = (np.frombuffer((C.c_char * (24 * no.value)).from_address(bp), dtype=GBOUND_DTYPE).copy()
                                    if no.value else np.zeros(0, GBOUND_DTYPE))
            lp2 = L.str_er_result_line_labels(rh, C.byref(no))
            if lp2:
                k = no.value
                res.line_label = np.frombuffer((C.c_char * (4 * k)).from_address(lp2), dtype=np.int32).copy() if k else np.zeros(0, np.int32)
                pp2 = L.str_er_result_line_probs(rh, C.byref(no))
                res.line_prob = np.frombuffer((C.c_char * (8 * k)).from_address(pp2), dtype=np.float64).copy() if k else np.zeros(0, np.float64)
                kp2 = L.str_er_result_line_kept(rh, C.byref(no))
                res.line_kept = np.frombuffer((C.c_char * k).from_address(kp2), dtype=np.uint8).copy().astype(bool) if k else np.zeros(0, bool)
                ap2 = L.str_er_result_text_alive(rh, C.byref(no))
                res.text_alive = (np.frombuffer((C.c_char * no.value).from_address(ap2), dtype=np.uint8).copy().astype(bool)
                                  if no.value else np.zeros(0, bool))
            return res
        finally:
            L.str_er_result_free(rh)

    def last_tree_stats(self) -> dict:
        """Node records / border pixel pairs / tiles of the last detect call (str_er_last_tree_stats)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.L.str_er_last_tree_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"records": int(a.value), "seam_pairs": int(b.value), "tiles": int(c.value)}

    def tile2_stats(self) -> dict:
        """Tiles given to the second tile kernel (k_tile_tree2) since the context was created, and how many it handed back (str_er_tile2_stats)."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.L.str_er_tile2_stats(self.h, C.byref(a), C.byref(b)))
        return {"tiles": int(a.value), "handed_back": int(b.value)}

    def set_profiling(self, on: bool = True) -> None:
        """Per-kernel-group HIP events for the calls that follow (Result.profile / last_profile()); off by default: an event between two kernels costs
        stream time (str_er_set_profiling)."""
        self._check(self.L.str_er_set_profiling(self.h, 1 if on else 0))

    def ocr_stage_stats(self) -> dict:
        """Batches whose STAGE_OCR scores were computed right behind classify (sized from the previous batch), and batches scored again after the
        counters were read (str_er_ocr_stage_stats)."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.L.str_er_ocr_stage_stats(self.h, C.byref(a), C.byref(b)))
        return {"scored_early": int(a.value), "scored_again": int(b.value)}

    def last_profile(self) -> dict:
        names = (C.c_char_p * 32)()
        ms = (C.c_double * 32)()
        k = self.L.str_er_last_profile(self.h, names, ms, 32)
        return {names[i].decode(): ms[i] for i in range(min(k, 32))}

    # ---- the hot path ---------------------------------------------------------------------------
    def text_detect(self, src: np.ndarray, stages: int = STAGE_ALL, want_nodes: bool = False) -> Result:
        """ERFilter::text_detect up to classify (src/ER.cpp:33-60) for one BGR frame (H,W,3)
        or a batch (F,H,W,3) of uint8."""
        a = np.ascontiguousarray(src, dtype=np.uint8)
        if a.ndim == 3:
            a = a[None]
        if a.ndim != 4 or a.shape[3] != 3:
            raise ValueError("expected (H,W,3) or (F,H,W,3) uint8 BGR")
        f, h, w, _ = a.shape
        rh = C.c_void_p()
        self._check(self.L.str_er_detect_bgr(self.h, _np_ptr(a), w, h, 3 * w, 3 * w * h, f, MEM_HOST,
                                             stages | (WANT_NODES if want_nodes else 0), C.byref(rh)))
        return self._collect(rh)

    def text_detect_nv12(self, nv12: np.ndarray, w: int, h: int, stages: int = STAGE_ALL, want_nodes: bool = False) -> Result:
        """str_er_detect_nv12: frames as a video decoder delivers them, (h * 3 // 2, w) or (F, h * 3 // 2, w) uint8 (luma plane,
        then interleaved Cb/Cr at half resolution); the NV12 -> Y/Cr/Cb step is build-defined (include/str_er.h)."""
        a = np.ascontiguousarray(nv12, dtype=np.uint8)
        if a.ndim == 2:
            a = a[None]
        f = a.shape[0]
        assert a.shape[1:] == (h + h // 2, w), a.shape
        rh = C.c_void_p()
        self._check(self.L.str_er_detect_nv12(self.h, _np_ptr(a), w, h, w, w * (h + h // 2), f, MEM_HOST,
                                              stages | (WANT_NODES if want_nodes else 0), C.byref(rh)))
        return self._collect(rh)

    def text_detect_planes(self, src: np.ndarray, select, stages: int = STAGE_ALL, want_nodes: bool = False) -> Result:
        """text_detect for a subset of the logical planes (str_er_detect_bgr_planes): select[level * n_channels + k] flags."""
        a = np.ascontiguousarray(src, dtype=np.uint8)
        if a.ndim == 3:
            a = a[None]
        f, h, w, _ = a.shape
        sel = np.ascontiguousarray(select, dtype=np.uint8)
        rh = C.c_void_p()
        self._check(self.L.str_er_detect_bgr_planes(self.h, _np_ptr(a), w, h, 3 * w, 3 * w * h, f, MEM_HOST,
                                                    stages | (WANT_NODES if want_nodes else 0), _np_ptr(sel), len(sel), C.byref(rh)))
        return self._collect(rh)

    # ---- the level-0 planes of one frame in strips over several GPUs ---------------------------
    def strip_extract(self, frame: np.ndarray, strip: int, n_strips: int) -> bytes:
        """Tile trees of strip `strip` of `n_strips` of every channel's level-0 plane: the bytes to send to the plane's owner."""
        a = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, _ = a.shape
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.L.str_er_strip_extract(self.h, _np_ptr(a), w, h, 3 * w, MEM_HOST, strip, n_strips, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p.value, n.value)
        finally:
            self.L.str_er_strip_free(p)

    def strip_merge(self, frame: np.ndarray, blobs, stages: int = STAGE_ALL, want_nodes: bool = False) -> Result:
        """The owner's half: all strips' blobs (in strip order) -> the result text_detect gives for the frame."""
        a = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, _ = a.shape
        k = len(blobs)
        bufs = [C.create_string_buffer(bytes(x), len(x)) for x in blobs]
        ptrs = (C.c_void_p * k)(*[C.cast(x, C.c_void_p) for x in bufs])
        sizes = (C.c_int64 * k)(*[len(x) for x in blobs])
        rh = C.c_void_p()
        self._check(self.L.str_er_strip_merge(self.h, _np_ptr(a), w, h, 3 * w, MEM_HOST, ptrs, sizes, k,
                                              stages | (WANT_NODES if want_nodes else 0), C.byref(rh)))
        return self._collect(rh)

    def strip_extract_dev(self, frame: np.ndarray, strip: int, n_strips: int):
        """The same, the blob left in a device buffer of this context (valid until its next strip call): (device address, bytes)."""
        a = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, _ = a.shape
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.L.str_er_strip_extract_dev(self.h, _np_ptr(a), w, h, 3 * w, MEM_HOST, strip, n_strips, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def strip_merge_ex(self, frame: np.ndarray, blobs, sizes=None, device_blobs: bool = False, plane_select=None, stages: int = STAGE_ALL,
                       want_nodes: bool = False) -> Result:
        """str_er_strip_merge_ex: blobs = bytes objects (host) or device addresses with `sizes` (device_blobs=True); plane_select =
        one flag per channel of the context: the channels this owner puts together (None: all)."""
        a = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, _ = a.shape
        k = len(blobs)
        if device_blobs:
            ptrs = (C.c_void_p * k)(*[C.c_void_p(int(x)) for x in blobs])
            szs = (C.c_int64 * k)(*[int(x) for x in sizes])
        else:
            bufs = [C.create_string_buffer(bytes(x), len(x)) for x in blobs]
            ptrs = (C.c_void_p * k)(*[C.cast(x, C.c_void_p) for x in bufs])
            szs = (C.c_int64 * k)(*[len(x) for x in blobs])
        sel = np.ascontiguousarray(plane_select, dtype=np.uint8) if plane_select is not None else None
        rh = C.c_void_p()
        self._check(self.L.str_er_strip_merge_ex(self.h, _np_ptr(a), w, h, 3 * w, MEM_HOST, ptrs, szs, MEM_DEVICE if device_blobs else MEM_HOST, k,
                                                 _np_ptr(sel) if sel is not None else None, stages | (WANT_NODES if want_nodes else 0), C.byref(rh)))
        return self._collect(rh)

    def detect_bgr_device(self, dptr: int, w: int, h: int, n_frames: int, stages: int = STAGE_ALL,
                          stride: Optional[int] = None, frame_pitch: Optional[int] = None) -> Result:
        """Same, for frames already resident in HBM (dptr = device address)."""
        stride = stride or 3 * w
        frame_pitch = frame_pitch or stride * h
        rh = C.c_void_p()
        self._check(self.L.str_er_detect_bgr(self.h, dptr, w, h, stride, frame_pitch, n_frames, MEM_DEVICE, stages,
                                             C.byref(rh)))
        return self._collect(rh)

    def detect_planes(self, planes: np.ndarray, stages: int = STAGE_ALL, want_nodes: bool = False) -> Result:
        """The loop body at src/ER.cpp:52-59 for (H,W) or (N,H,W) uint8 planes."""
        a = np.ascontiguousarray(planes, dtype=np.uint8)
        if a.ndim == 2:
            a = a[None]
        n, h, w = a.shape
        rh = C.c_void_p()
        self._check(self.L.str_er_detect_planes(self.h, _np_ptr(a), w, h, w, w * h, n, MEM_HOST,
                                                stages | (WANT_NODES if want_nodes else 0), C.byref(rh)))
        return self._collect(rh)

    def detect_planes_device(self, dptr: int, w: int, h: int, n_planes: int, stages: int = STAGE_ALL,
                             stride: Optional[int] = None, plane_pitch: Optional[int] = None) -> Result:
        stride = stride or w
        plane_pitch = plane_pitch or stride * h
        rh = C.c_void_p()
        self._check(self.L.str_er_detect_planes(self.h, dptr, w, h, stride, plane_pitch, n_planes, MEM_DEVICE, stages,
                                                C.byref(rh)))
        return self._collect(rh)

    # ---- single stages ---------------------------------------------------------------------------
    def compute_channels(self, src: np.ndarray) -> np.ndarray:
        """ERFilter::compute_channels (src/ER.cpp:114-128) -> (6,H,W) uint8."""
        a = np.ascontiguousarray(src, dtype=np.uint8)
        h, w, _ = a.shape
        out = np.empty((6, h, w), np.uint8)
        self._check(self.L.str_er_compute_channels(self.h, _np_ptr(a), w, h, 3 * w, _np_ptr(out)))
        return out

    def er_tree_extract(self, plane: np.ndarray) -> PlaneResult:
        """ERFilter::er_tree_extract (src/ER.cpp:240-374): the kept tree as a node table."""
        return self.detect_planes(plane, STAGE_EXTRACT, want_nodes=True).planes[0]

    def non_maximum_supression(self, nodes: np.ndarray, rows: int, cols: int, plane: Optional[np.ndarray] = None):
        """ERFilter::non_maximum_supression (src/ER.cpp:416-505) on a node table.
        Returns (pool indices in ascending key order, ambiguous count).  Sibling ties (sibling_order = 0): without `plane` the
        table order is the child-list order; with the (rows, cols) uint8 plane they are decided by replaying the reference's flood."""
        nd = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        cap = max(1, len(nd))
        pool = np.zeros(cap, np.int32)
        n, amb = C.c_int32(), C.c_int32()
        if plane is None:
            self._check(self.L.str_er_nms_tree(self.h, _np_ptr(nd), len(nd), rows, cols, _np_ptr(pool), cap, C.byref(n),
                                               C.byref(amb)))
        else:
            a = np.ascontiguousarray(plane, dtype=np.uint8)
            if a.shape != (rows, cols):
                raise ValueError("plane must be (rows, cols)")
            self._check(self.L.str_er_nms_tree_plane(self.h, _np_ptr(nd), len(nd), _np_ptr(a), cols, rows, cols, _np_ptr(pool), cap,
                                                     C.byref(n), C.byref(amb)))
        return pool[:n.value].copy(), amb.value

    def classify(self, plane: np.ndarray, boxes_xywh: np.ndarray):
        """ERFilter::classify (src/ER.cpp:507-528): (cls, score_strong, score_weak) per box."""
        a = np.ascontiguousarray(plane, dtype=np.uint8)
        b = np.ascontiguousarray(boxes_xywh, dtype=np.int32).reshape(-1, 4)
        n = len(b)
        cls = np.zeros(n, np.uint8)
        ss = np.zeros(n, np.float64)
        sw = np.zeros(n, np.float64)
        self._check(self.L.str_er_classify_boxes(self.h, _np_ptr(a), a.shape[1], a.shape[0], a.shape[1], _np_ptr(b), n,
                                                 _np_ptr(cls), _np_ptr(ss), _np_ptr(sw)))
        return cls, ss, sw

    def predict(self, which: int, fv: np.ndarray) -> np.ndarray:
        """stc->predict(fv) / wtc->predict(fv) (inc/adaboost.h:131) for (n,1024) feature vectors."""
        a = np.ascontiguousarray(fv, dtype=np.float64).reshape(-1, 1024)
        out = np.zeros(len(a), np.float64)
        self._check(self.L.str_er_cascade_predict(self.h, which, _np_ptr(a), len(a), _np_ptr(out)))
        return out

    # ---- OCR scorer, SVM half (config 3): OCR::OCR loads the model, chain_run calls svm_predict_probability ----
    def load_svm_model(self, path: str, dim: int = 1800) -> None:
        self._check(self.L.str_er_load_svm_model(self.h, path.encode(), dim))

    def load_svm_model_text(self, text: bytes, dim: int = 1800) -> None:
        self._check(self.L.str_er_load_svm_model_mem(self.h, text, len(text), dim))

    def svm_info(self):
        a, b, d = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.L.str_er_svm_info(self.h, C.byref(a), C.byref(b), C.byref(d)))
        return a.value, b.value, d.value

    def svm_forms(self) -> dict:
        """How the loaded model is evaluated (str_er_svm_forms): support vectors as bytes (exact 8-bit kernel matrix) or as three bf16 pieces; decision values
        summed per class (f64 matrix products) or per vector."""
        a, b = C.c_int32(), C.c_int32()
        self._check(self.L.str_er_svm_forms(self.h, C.byref(a), C.byref(b)))
        return {"bytes": bool(a.value), "class_sums": bool(b.value)}

    def svm_predict_probability(self, x: np.ndarray, want_dec: bool = False):
        """svm_predict_probability (src/svm.cpp:2592-2629) for dense (n, dim) features -> (label, prob[, dec])."""
        a = np.ascontiguousarray(x, dtype=np.float64)
        if a.ndim == 1:
            a = a[None]
        n, dim = a.shape
        k = self.svm_info()[0]
        label = np.zeros(n, np.int32)
        prob = np.zeros((n, k), np.float64)
        dec = np.zeros((n, k * (k - 1) // 2), np.float64) if want_dec else None
        self._check(self.L.str_er_svm_predict_probability(self.h, _np_ptr(a), n, dim, _np_ptr(label), _np_ptr(prob),
                                                          _np_ptr(dec) if want_dec else None))
        return (label, prob, dec) if want_dec else (label, prob)

    def chain_run(self, plane: np.ndarray, boxes_xywh: np.ndarray, classify: bool = True, slope=None):
        """OCR::chain_run (src/OCR.cpp:67-140) for every box: (q[n,1800] uint8, label, prob) or q only.
        slope: None (all 0), one number (the Text line's slope, src/ER.cpp:731) or one per box."""
        a = np.ascontiguousarray(plane, dtype=np.uint8)
        b = np.ascontiguousarray(boxes_xywh, dtype=np.int32).reshape(-1, 4)
        n = len(b)
        q = np.zeros((n, 1800), np.uint8)
        label = np.zeros(n, np.int32)
        prob = np.zeros(n, np.float64)
        sl = None
        if slope is not None:
            sl = np.ascontiguousarray(np.broadcast_to(np.asarray(slope, np.float64), (n,)))
        self._check(self.L.str_er_ocr_chain_run_slope(self.h, _np_ptr(a), a.shape[1], a.shape[0], a.shape[1], _np_ptr(b),
                                                      _np_ptr(sl) if sl is not None else None, n,
                                                      _np_ptr(label) if classify else None, _np_ptr(prob) if classify else None, _np_ptr(q)))
        return (q, label, prob) if classify else q

    def set_min_ocr_prob(self, p: float) -> None:
        """MIN_OCR_PROB, ERFilter's last constructor argument (inc/ER.h:113)."""
        self._check(self.L.str_er_set_min_ocr_prob(self.h, float(p)))
        self.min_ocr_prob = float(p)

    # ---- the first consumers of the classified ERs --------------------------------
    def calc_color(self, mask_plane: np.ndarray, color_img: np.ndarray, boxes_xywh: np.ndarray) -> np.ndarray:
        """calc_color (src/ER.cpp:1391-1419) for every box: [n,3] = ER::color1..3 (color_img: H x W x 3 uint8, the Ycrcb Mat)."""
        a = np.ascontiguousarray(mask_plane, dtype=np.uint8)
        ci = np.ascontiguousarray(color_img, dtype=np.uint8)
        if ci.ndim != 3 or ci.shape[2] != 3:
            raise ValueError("color_img must be H x W x 3")
        b = np.ascontiguousarray(boxes_xywh, dtype=np.int32).reshape(-1, 4)
        out = np.zeros((len(b), 3), np.float64)
        self._check(self.L.str_er_calc_color(self.h, _np_ptr(a), a.shape[1], a.shape[0], a.shape[1], _np_ptr(ci), ci.shape[1], ci.shape[0],
                                             ci.shape[1] * 3, _np_ptr(b), len(b), _np_ptr(out)))
        return out

    def er_track(self, cands: np.ndarray, colors: np.ndarray):
        """ERFilter::er_track (src/ER.cpp:530-590) on the ERs of one image: cands (CAND_DTYPE; cls 1 = strong[], 2 = weak[]) and
        their colours [n,3] -> (tracked[n] bool, cx[n], cy[n])."""
        cd = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        col = np.ascontiguousarray(colors, dtype=np.float64).reshape(-1, 3)
        n = len(cd)
        if len(col) != n:
            raise ValueError("one colour triple per candidate")
        tr = np.zeros(n, np.uint8)
        cx = np.zeros(n, np.int32)
        cy = np.zeros(n, np.int32)
        self._check(self.L.str_er_er_track(self.h, _np_ptr(cd), _np_ptr(col), n, _np_ptr(tr), _np_ptr(cx), _np_ptr(cy)))
        return tr.astype(bool), cx, cy

    def er_grouping(self, cands: np.ndarray, tracks: np.ndarray, overlap_sup: bool = False, inner_sup: bool = False) -> Result:
        """ERFilter::er_grouping(all_er, text, overlap_sup, inner_sup) (src/ER.cpp:612-692) on the ERs of one image
        (cands CAND_DTYPE, tracks TRACK_DTYPE with color1-3 / cx / cy / tracked): Result with .texts / .text_ers / .group_bounds."""
        cd = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        tr = np.ascontiguousarray(tracks, dtype=TRACK_DTYPE)
        if len(cd) != len(tr):
            raise ValueError("one track record per candidate")
        rh = C.c_void_p()
        self._check(self.L.str_er_er_grouping(self.h, _np_ptr(cd), _np_ptr(tr), len(cd), int(overlap_sup), int(inner_sup), C.byref(rh)))
        return self._collect(rh)

    def make_LBP_hist(self, plane: np.ndarray, boxes_xywh: Optional[np.ndarray] = None, return_tiles: bool = False):
        """ERFilter::make_LBP_hist(input, 2, 24) (src/ER.cpp:789-816).  With no boxes the whole
        plane is the ROI (what the reference's get_lbp_data does, src/utils.cpp:1451-1470)."""
        a = np.ascontiguousarray(plane, dtype=np.uint8)
        if boxes_xywh is None:
            boxes_xywh = np.array([[0, 0, a.shape[1], a.shape[0]]], np.int32)
        b = np.ascontiguousarray(boxes_xywh, dtype=np.int32).reshape(-1, 4)
        n = len(b)
        hist = np.zeros((n, 1024), np.float64)
        tiles = np.zeros((n, 26, 26), np.uint8) if return_tiles else None
        self._check(self.L.str_er_lbp_hist(self.h, _np_ptr(a), a.shape[1], a.shape[0], a.shape[1], _np_ptr(b), n,
                                           _np_ptr(hist), _np_ptr(tiles) if return_tiles else None))
        return (hist, tiles) if return_tiles else hist

    def calc_LBP(self, plane: np.ndarray, boxes_xywh: Optional[np.ndarray] = None) -> np.ndarray:
        """ERFilter::calc_LBP(input, 24) (inc/ER.h:134, src/ER.cpp:819-845): the 24x24 Mean-LBP code map of every box
        (of the whole plane with no boxes, which is how OCR::lbp_run calls it, src/OCR.cpp:37-39)."""
        a = np.ascontiguousarray(plane, dtype=np.uint8)
        if boxes_xywh is None:
            boxes_xywh = np.array([[0, 0, a.shape[1], a.shape[0]]], np.int32)
        b = np.ascontiguousarray(boxes_xywh, dtype=np.int32).reshape(-1, 4)
        out = np.zeros((len(b), 24, 24), np.uint8)
        self._check(self.L.str_er_calc_lbp(self.h, _np_ptr(a), a.shape[1], a.shape[0], a.shape[1], _np_ptr(b), len(b), _np_ptr(out)))
        return out

    def resize_plane(self, src: np.ndarray, dw: int, dh: int) -> np.ndarray:
        a = np.ascontiguousarray(src, dtype=np.uint8)
        out = np.empty((dh, dw), np.uint8)
        self._check(self.L.str_er_resize_plane(self.h, _np_ptr(a), a.shape[1], a.shape[0], a.shape[1], _np_ptr(out), dw, dh))
        return out

    def tie_stats(self) -> dict:
        """Exact NMS ties: planes whose flood order had to be walked on a host core so far, the host time that took, pool size."""
        n, ms, th = C.c_uint64(), C.c_double(), C.c_int32()
        self._check(self.L.str_er_tie_stats(self.h, C.byref(n), C.byref(ms), C.byref(th)))
        return {"planes_walked": int(n.value), "walk_ms_total": float(ms.value), "host_threads": int(th.value)}

    def workspace_bytes(self) -> int:
        return int(self.L.str_er_workspace_bytes(self.h))


def flood_order(plane: np.ndarray, thresh_step: int = 8) -> np.ndarray:
    """str_er_flood_order: (h, w) uint32, 1-based order in which the reference's flood first reaches each pixel (0: never)."""
    a = np.ascontiguousarray(plane, dtype=np.uint8)
    h, w = a.shape
    out = np.zeros((h, w), np.uint32)
    rc = load_library().str_er_flood_order(_np_ptr(a), w, h, w, thresh_step, _np_ptr(out))
    if rc != 0:
        raise StrErError(rc, "str_er_flood_order")
    return out


class Comm:
    """One rank of the candidate gather (include/str_er.h, str_er_comm_* / str_er_gather_*): an RCCL communicator
    (`Comm.rccl`) or a member of an in-process group that exchanges through host memory (`Comm.local_group`)."""

    def __init__(self, handle, world: int, rank: int, group=None):
        self.L, self.h, self.world, self.rank, self._group = load_library(), handle, world, rank, group

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = load_library().str_er_comm_unique_id(buf)
        if rc != 0:
            raise StrErError(rc, "str_er_comm_unique_id (is librccl.so there?)")
        return buf.raw

    @classmethod
    def rccl(cls, device: int, rank: int, world: int, uid: bytes) -> "Comm":
        h = C.c_void_p()
        L = load_library()
        rc = L.str_er_comm_create(device, rank, world, C.create_string_buffer(uid, 128), C.byref(h))
        if rc != 0:
            raise StrErError(rc, "str_er_comm_create: " + (L.str_er_comm_last_error(None) or b"").decode())
        return cls(h, world, rank)

    @classmethod
    def local_group(cls, world: int):
        """`world` communicators of one in-process group (use one per thread)."""
        L = load_library()
        g = C.c_void_p()
        rc = L.str_er_comm_local_group(world, C.byref(g))
        if rc != 0:
            raise StrErError(rc, "str_er_comm_local_group")
        out = []
        for r in range(world):
            h = C.c_void_p()
            rc = L.str_er_comm_create_local(g, r, C.byref(h))
            if rc != 0:
                raise StrErError(rc, "str_er_comm_create_local")
            out.append(cls(h, world, r, group=g))
        L.str_er_comm_local_group_free(g)        # the communicators keep the group alive
        return out

    def _take(self, rc, p, n, counts):
        if rc != 0:
            raise StrErError(rc, (self.L.str_er_comm_last_error(self.h) or b"").decode())
        try:
            out = (np.frombuffer((C.c_char * (48 * n.value)).from_address(p.value), dtype=CAND_DTYPE).copy()
                   if n.value else np.zeros(0, CAND_DTYPE))
        finally:
            self.L.str_er_gather_free(p)
        return out, np.array(counts[:], np.int32)

    def gather(self, cands: np.ndarray, frame_offset: int = 0, failed: bool = False):
        """Collective: (all ranks' records ordered by rank, per-rank counts); `frame_offset` is added to this rank's frames.
        failed=True: this rank has nothing valid to contribute -- it still takes part (a count of -1), and EVERY rank's call returns an
        error instead of one rank leaving the others waiting in the collective."""
        a = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        p, n = C.c_void_p(), C.c_int32()
        counts = (C.c_int32 * self.world)()
        rc = self.L.str_er_gather_cands(self.h, _np_ptr(a) if len(a) and not failed else None, -1 if failed else len(a), frame_offset,
                                        C.byref(p), C.byref(n), counts)
        return self._take(rc, p, n, counts)

    def gather_last(self, erf: "ERFilter", frame_offset: int = 0):
        """The same for the candidates of erf's last detect call, straight from the device array (RCCL only)."""
        p, n = C.c_void_p(), C.c_int32()
        counts = (C.c_int32 * self.world)()
        rc = self.L.str_er_gather_last(self.h, erf.h, frame_offset, C.byref(p), C.byref(n), counts)
        return self._take(rc, p, n, counts)

    def allgather_bytes(self, data, device_in: bool = False, device_out: bool = False):
        """Collective, variable length (str_er_comm_allgather_bytes).  data: bytes / uint8 array, or (device address, n) with
        device_in.  Host output: list of `world` bytes objects; device output: (base address, starts, sizes) -- the communicator's
        buffer, valid until its next collective."""
        if device_in:
            ptr, n = C.c_void_p(int(data[0])), int(data[1])
            keep = None
        else:
            keep = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8).reshape(-1)
            ptr, n = (_np_ptr(keep) if len(keep) else None), len(keep)
        out = C.c_void_p()
        starts, sizes = (C.c_int64 * self.world)(), (C.c_int64 * self.world)()
        rc = self.L.str_er_comm_allgather_bytes(self.h, ptr, n, MEM_DEVICE if device_in else MEM_HOST, MEM_DEVICE if device_out else MEM_HOST,
                                                C.byref(out), starts, sizes)
        if rc != 0:
            raise StrErError(rc, (self.L.str_er_comm_last_error(self.h) or b"").decode())
        if device_out:
            return int(out.value or 0), [int(v) for v in starts], [int(v) for v in sizes]
        try:
            return [C.string_at(out.value + starts[r], sizes[r]) if sizes[r] else b"" for r in range(self.world)]
        finally:
            self.L.str_er_comm_free(out)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.L.str_er_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrameStream:
    """Frame ingest (include/str_er.h, str_er_stream_*): `depth` batches of host frames in flight, uploads overlapping
    compute.  acquire() hands out a page-locked numpy buffer to decode into; results come back in submission order."""

    def __init__(self, params: Params, depth: int = 3):
        self.L = load_library()
        p = params
        cp = _Params(p.thresh_step, p.min_area, p.max_area, p.stability_t, p.overlap_coef, p.n_pyr_levels,
                     p.channel_mask, p.device, p.max_width, p.max_height, p.max_frames, p.kept_cap, p.pool_cap,
                     p.sibling_order, p.stream)
        h = C.c_void_p()
        rc = self.L.str_er_stream_create(C.byref(cp), depth, C.byref(h))
        if rc != 0:
            raise StrErError(rc, (self.L.str_er_last_error(None) or b"").decode())
        self.h = h
        self.params = p

    def close(self) -> None:
        if getattr(self, "h", None):
            self.L.str_er_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise StrErError(rc, (self.L.str_er_stream_last_error(self.h) or b"").decode())

    @property
    def depth(self) -> int:
        return int(self.L.str_er_stream_depth(self.h))

    def load_cascade(self, which: int, path: str) -> None:
        self._check(self.L.str_er_stream_load_cascade(self.h, which, path.encode()))

    def acquire(self):
        """(slot, uint8 view of the pinned staging buffer)."""
        slot, buf, cap = C.c_int32(), C.c_void_p(), C.c_int64()
        self._check(self.L.str_er_stream_acquire(self.h, C.byref(slot), C.byref(buf), C.byref(cap)))
        arr = np.frombuffer((C.c_uint8 * cap.value).from_address(buf.value), dtype=np.uint8)
        return slot.value, arr

    def submit(self, slot: int, w: int, h: int, n_frames: int, stages: int = STAGE_ALL) -> int:
        t = C.c_uint64()
        self._check(self.L.str_er_stream_submit(self.h, slot, w, h, 3 * w, 3 * w * h, n_frames, stages, C.byref(t)))
        return int(t.value)

    def submit_nv12(self, slot: int, w: int, h: int, n_frames: int, stages: int = STAGE_ALL) -> int:
        """The staging buffer holds n_frames tightly packed NV12 frames (w * h * 3 / 2 bytes each)."""
        t = C.c_uint64()
        self._check(self.L.str_er_stream_submit_nv12(self.h, slot, w, h, w, w * (h + h // 2), n_frames, stages, C.byref(t)))
        return int(t.value)

    def submit_copy(self, frames: np.ndarray, stages: int = STAGE_ALL) -> int:
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        if a.ndim == 3:
            a = a[None]
        n, h, w, _ = a.shape
        t = C.c_uint64()
        self._check(self.L.str_er_stream_submit_copy(self.h, _np_ptr(a), w, h, 3 * w, 3 * w * h, n, stages, C.byref(t)))
        return int(t.value)

    def pending(self) -> int:
        return int(self.L.str_er_stream_pending(self.h))

    def next(self):
        """(ticket, Result) of the oldest submitted batch; blocks until it is done."""
        rh, t = C.c_void_p(), C.c_uint64()
        rc = self.L.str_er_stream_next(self.h, C.byref(rh), C.byref(t))
        self._check(rc)
        shim = object.__new__(ERFilter)
        shim.L = self.L
        shim.h = None
        return int(t.value), ERFilter._collect(shim, rh, profile={})
