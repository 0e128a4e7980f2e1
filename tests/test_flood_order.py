"""The walk that decides NMS sibling ties (str_er_flood_order = flood_order.cpp, the product's statement of the ORDER of the
reference's flood, src/ER.cpp:240-374) against the oracle's restatement of the flood itself: the oracle builds its child lists
the way the reference does (children prepended when merged, src/ER.cpp:183-185), so along every child list the stamps of the
children's key pixels must be strictly decreasing -- first in the list = entered last.  Runs on the CPU (no GPU needed)."""
import numpy as np
import pytest


def _planes(S):
    rng = np.random.default_rng(5)
    yield 8, rng.integers(0, 256, (60, 90), dtype=np.uint8)
    yield 8, (rng.integers(0, 4, (70, 50)) * 60 + rng.integers(0, 9, (70, 50))).astype(np.uint8)
    yield 8, S.synth.gray(S.synth.stext_bgr(3, 200, 150))
    yield 8, rng.integers(200, 256, (40, 64), dtype=np.uint8)                 # sentinel walls everywhere
    yield 2, rng.integers(0, 256, (50, 70), dtype=np.uint8)
    yield 4, rng.integers(0, 256, (50, 70), dtype=np.uint8)                   # sentinel level 64: the last one that needs the two-byte state
    yield 5, rng.integers(0, 256, (50, 70), dtype=np.uint8)                   # sentinel level 52: one byte of state
    yield 1, rng.integers(0, 256, (33, 47), dtype=np.uint8)                   # 256 levels, no sentinel
    yield 16, np.kron(rng.integers(0, 256, (12, 20), dtype=np.uint8), np.ones((5, 5), np.uint8))
    wall = rng.integers(0, 250, (30, 30), dtype=np.uint8)
    wall[0, 0] = 255                                                            # start pixel at the sentinel level: 1, then w
    yield 8, wall
    wall2 = wall.copy(); wall2[0, 1] = 255
    yield 8, wall2
    yield 8, np.full((7, 9), 100, np.uint8)
    yield 8, np.full((1, 1), 3, np.uint8)


def test_flood_order_matches_the_oracles_child_lists(S, oracle):
    n_lists = 0
    for step, img in _planes(S):
        st = S.flood_order(img, step).reshape(-1)
        t = oracle.tree_extract(img, step, 0)              # MIN_AREA 0: every node is kept, every sibling order is visible
        nodes = t.nodes
        assert st[int(nodes[t.root]["key"])] > 0
        for p in range(len(nodes)):
            c = int(nodes[p]["child"])
            prev = None
            while c >= 0:
                s = int(st[int(nodes[c]["key"])])
                assert s > 0, "a pixel of the tree was not reached"
                if prev is not None:
                    assert s < prev, (step, img.shape, p)
                    n_lists += 1
                prev = s
                c = int(nodes[c]["next"])
    assert n_lists > 1000


def test_flood_order_is_a_permutation_of_the_reached_pixels(S, oracle):
    for step, img in _planes(S):
        st = S.flood_order(img, step).reshape(-1)
        reached = np.sort(st[st > 0])
        assert (reached == np.arange(1, len(reached) + 1)).all()
        # the tree covers exactly the reached pixels that are below the sentinel level; the reached sentinel
        # pixels are marked but never flooded
        t = oracle.tree_extract(img, step, 0)
        hi = 255 // step + 1
        lv = oracle.quantise(img, step).reshape(-1) if hasattr(oracle, "quantise") else None
        if lv is not None and int(t.nodes[t.root]["level"]) < hi:
            assert int(t.nodes[t.root]["npix"]) == int(((st > 0) & (lv < hi)).sum())


def test_flood_order_arguments(S):
    import ctypes as C
    L = S.load_library()
    out = np.zeros(4, np.uint32)
    img = np.zeros((2, 2), np.uint8)
    assert L.str_er_flood_order(img.ctypes.data, 2, 2, 1, 8, out.ctypes.data) == -1      # stride < w
    assert L.str_er_flood_order(img.ctypes.data, 2, 2, 2, 0, out.ctypes.data) == -1
    assert L.str_er_flood_order(None, 2, 2, 2, 8, out.ctypes.data) == -1
    assert S.flood_order(img).tolist() == [[1, 2], [3, 4]] or S.flood_order(img).max() == 4
