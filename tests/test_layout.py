"""The oracle is test infrastructure: nothing in the product may import, link or call it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "scene-text-recognition_amd")


def _product_files():
    for base in (PKG, os.path.join(ROOT, "include")):
        for dp, _, fns in os.walk(base):
            for fn in fns:
                if fn.endswith((".py", ".cpp", ".hip", ".inl", ".h", ".hpp")):
                    yield os.path.join(dp, fn)
    yield os.path.join(ROOT, "str_er_amd.py")


def test_product_never_touches_the_oracle():
    pat = re.compile(r"(from\s+oracle|import\s+oracle|er_oracle|liber_oracle|libref_adaboost|oracle/)")
    bad = []
    for p in _product_files():
        for i, line in enumerate(open(p, errors="replace"), 1):
            if pat.search(line) and "see oracle/er_oracle.c for the" not in line:
                bad.append(f"{p}:{i}: {line.strip()}")
    assert not bad, "\n".join(bad)


def test_no_reference_path_at_run_time():
    """/root/reference does not exist on the GPU box: product, bench and GPU tests must not read it."""
    bad = []
    files = list(_product_files()) + [os.path.join(ROOT, "bench.py")]
    files += [os.path.join(ROOT, "tests", f) for f in os.listdir(os.path.join(ROOT, "tests")) if f.startswith("test_gpu")]
    for p in files:
        for i, line in enumerate(open(p, errors="replace"), 1):
            if re.search(r"""open\(|load\(|CDLL\(""", line) and "/root/reference" in line:
                bad.append(f"{p}:{i}")
    assert not bad, bad


def test_no_compat_layers():
    """No CUDA shims, no dual CUDA/HIP paths, no Triton."""
    for p in _product_files():
        txt = open(p, errors="replace").read()
        assert "__HIP_PLATFORM_AMD__" not in txt and "import triton" not in txt and "cuda_runtime.h" not in txt, p


def test_repo_layout():
    for rel in ("bench.py", "__graft_entry__.py", "include/str_er.h", "oracle/er_oracle.c", "oracle/Makefile",
                "scene-text-recognition_amd/data/cascades.npz", "tests/golden/make_cascades.py", "tests/golden/cascade_vectors.npz"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
