"""scene-text-recognition_amd -- MI355X (gfx950) extremal-region text-detection hot path.

Python binding (ctypes) of the C ABI in include/str_er.h.  The directory name carries a
hyphen (it follows the reference's repository name), so import it with

    import importlib; str_er = importlib.import_module("scene-text-recognition_amd")

or through the alias module `str_er_amd` at the repository root.

Everything here forwards to libstr_er_hip.so: there is no Python or CPU implementation
of the path, and loading fails loudly if the HIP library has not been built.
"""
from __future__ import annotations

from .binding import (  # noqa: F401
    CLS_POOL, CLS_STRONG, CLS_WEAK, STAGE_ALL, STAGE_CLASSIFY, STAGE_OCR, STAGE_EXTRACT, STAGE_NMS, STAGE_TRACK, STAGE_GROUP, GROUP_INNER_SUP, GROUP_OVERLAP_SUP, STAGE_OCR_LINES, TRACK_DTYPE, TEXT_DTYPE, GBOUND_DTYPE, WANT_NODES,
    CAND_DTYPE, NODE_DTYPE, PLANE_DTYPE, ERFilter, FrameStream, PlaneResult, Params, Result, StrErError, apply_runtime_hint, set_batch_slots, lib_path, load_library,
    flood_order, Comm,
)
from . import cascade_io, synth  # noqa: F401

__all__ = ["ERFilter", "Params", "Result", "StrErError", "load_library", "lib_path", "cascade_io", "synth",
           "CAND_DTYPE", "NODE_DTYPE"]


def __getattr__(name):
    # `dist` pulls in torch; load it only when somebody asks for it
    if name == "dist":
        import importlib
        return importlib.import_module(__name__ + ".dist")
    raise AttributeError(name)
