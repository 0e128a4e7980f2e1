#!/usr/bin/env python3
"""Build libstr_er_hip.so (gfx950) in-tree with hipcc.

    python scene-text-recognition_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The library is written for MI355X (gfx950) only.
-ffp-contract=off keeps the resize coefficient arithmetic identical to the host statement
of cv::resize (no fused multiply-add where the reference has a separate multiply and add).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libstr_er_hip.so")
SOURCES = ["er_kernels.hip", "ocr_kernels.hip", "track_kernels.hip", "er_group.cpp", "flood_order.cpp", "gather.cpp", "str_er_api.cpp", "api_models.cpp", "api_strips.cpp", "api_stages.cpp", "stream_api.cpp"]
DEPS = SOURCES + ["er_planes.inl", "er_tile_tree.inl", "er_tile_tree2.inl", "tile2_body.h", "er_tree_passes.inl", "er_nms.inl", "er_classify.inl", "er_kernels.h", "ocr_kernels.h", "ocr_device.h", "svm_tables.h", "str_er_ctx.h", "track_kernels.h", "er_device.h", "er_group.h", "flood_order.h", "er_types.h", os.path.join("..", "..", "include", "str_er.h")]
EXTRA = os.environ.get("STR_ER_EXTRA_FLAGS", "").split()
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-bitwise-instead-of-logical"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS) or os.path.getmtime(__file__) > t


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", os.path.splitext(src)[0] + ".o")
        cmd = [hipcc(), *FLAGS, *EXTRA, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
