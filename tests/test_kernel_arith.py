"""The integer / floating-point shortcuts the round-4 kernels take, restated in numpy and checked against the plain forms they replace.

These are host-side restatements of device arithmetic (er_kernels.hip: k_nms's overlap test, k_classify's reciprocal divisions, packed histogram and LBP
bits, k_resize's 24-bit products and row walk): the GPU parity tests prove the kernels' results, these pin WHY the shortcuts are exact, over their whole
input ranges, on a machine without a GPU.
"""
import math

import numpy as np
import pytest


# ---- k_nms: (double)as / (double)ap > coef decided from the sign of as - coef * ap unless that is within 1e-9 * ap of 0 ------------------------------

def _ratio_gt_shortcut(a, p, coef):
    x, y = np.float64(a), np.float64(p)
    d = x - np.float64(coef) * y
    if abs(d) > 1e-9 * y:
        return bool(d > 0.0)
    return bool(x / y > np.float64(coef))


@pytest.mark.parametrize("coef", [0.7, 0.2, 0.25, 0.5, 0.9, 0.3, 1.0 / 3.0, 0.6999999999999999, 0.7000000000000001])
def test_overlap_test_without_the_division(coef):
    rng = np.random.default_rng(int(coef * 1e6) & 0xFFFF)
    # box areas of a 1920 x 1080 plane and of a 3840 x 2160 one; the adversarial cases sit next to coef * ap
    for ap in np.concatenate([rng.integers(1, 1920 * 1080 + 1, 3000), rng.integers(1, 3840 * 2160 + 1, 1000), np.arange(1, 400)]):
        ap = int(ap)
        mid = int(math.floor(coef * ap))
        for a in {max(1, mid - 1), max(1, mid), mid + 1, mid + 2, int(rng.integers(1, ap + 1))}:
            if a > ap:
                continue
            want = bool(np.float64(a) / np.float64(ap) > np.float64(coef))
            assert _ratio_gt_shortcut(a, ap, coef) == want, (a, ap, coef)


def test_overlap_test_on_exact_quotients():
    # quotients that ARE the coefficient (2/10 against 0.2 ...): the shortcut must fall through to the division
    for num, den, coef in [(2, 10, 0.2), (1, 4, 0.25), (7, 10, 0.7), (9, 10, 0.9), (1, 2, 0.5), (3, 10, 0.3)]:
        for k in range(1, 2000, 7):
            a, p = num * k, den * k
            assert _ratio_gt_shortcut(a, p, coef) == bool(np.float64(a) / np.float64(p) > np.float64(coef))


# ---- k_classify: divisions by multiplication ----------------------------------------------------------------------------------------------------------

def test_reciprocal_division_by_the_tile_width():
    # ii / dw for ii < dw * dh <= 676, dw <= 26: (ii * ceil(65536 / dw)) >> 16
    for dw in range(1, 27):
        rcp = (65536 + dw - 1) // dw
        ii = np.arange(0, 26 * 26, dtype=np.uint32)
        assert rcp < (1 << 24) and int(ii.max()) * rcp < (1 << 32)          # v_mul_u32_u24's operands, and its 32-bit result
        np.testing.assert_array_equal((ii * np.uint32(rcp)) >> 16, ii // dw)


def test_division_by_24_of_the_lbp_index():
    idx = np.arange(0, 24 * 24, dtype=np.uint32)
    np.testing.assert_array_equal((idx * np.uint32(2731)) >> 16, idx // 24)


def test_lbp_bits_are_the_sign_of_sum_minus_8v():
    rng = np.random.default_rng(7)
    v = rng.integers(0, 256, (20000, 8)).astype(np.int64)
    v[:2000] = rng.integers(100, 104, (2000, 8))            # near-flat neighbourhoods: many 8 v == sum cases
    s = v.sum(axis=1)
    want = np.zeros(len(v), np.uint32)
    for k in range(8):
        want |= ((8 * v[:, k] > s).astype(np.uint32) << k)
    code = np.zeros(len(v), np.uint32)
    for k in range(7, -1, -1):                              # v_alignbit(code, diff, 31) = (code << 1) | (diff >> 31), v7 first
        diff = (s - 8 * v[:, k]).astype(np.int32).view(np.uint32)
        code = (code << np.uint32(1)) | (diff >> np.uint32(31))
    np.testing.assert_array_equal(code, want)


def test_histogram_counted_into_packed_bytes_never_carries():
    # a 24 x 24 LBP image has four 12 x 12 cells: at most 144 pixels fall into one bin -- a byte holds it, no carry into the neighbouring bin
    rng = np.random.default_rng(11)
    for trial in range(50):
        codes = rng.integers(0, 256, (24, 24)) if trial else np.zeros((24, 24), np.int64)       # (all pixels one code: the worst case)
        row = np.zeros(256, np.uint32)                       # the ER's packed row: 1024 bins, a byte each
        hist = np.zeros(1024, np.uint32)
        for i in range(24):
            for j in range(24):
                b = (512 if i >= 12 else 0) + (256 if j >= 12 else 0) + int(codes[i, j])
                hist[b] += 1
                row[b >> 2] += np.uint32(1 << (8 * (b & 3)))
        assert hist.max() <= 144
        np.testing.assert_array_equal(row.view(np.uint8), hist.astype(np.uint8))


# ---- cv::resize's fixed-point bilinear: every product fits 24 bits, nothing is negative ----------------------------------------------------------------

def test_resize_products_fit_24_bits():
    f = np.linspace(0.0, 1.0, 100001, dtype=np.float32)[:-1]
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    assert a0.min() >= 0 and a1.min() >= 0 and a0.max() <= 2048 and a1.max() <= 2048
    assert (a0 + a1).max() <= 2049                          # (the two roundings can add up to 2049, never more)
    h = 255 * (a0 + a1)                                     # horizontal sum of two taps
    assert h.max() < (1 << 24)
    assert 2048 < (1 << 24) and (h >> 4).max() < (1 << 24) and (2048 * (h >> 4)).max() < (1 << 32)


def _row_table(dy, scale_y, sh):
    fy = np.float32((dy + 0.5) * scale_y - 0.5)
    sy = int(math.floor(fy))
    return min(max(sy, 0), sh - 1), min(max(sy + 1, 0), sh - 1)


def test_resize_rows_are_adjacent_or_equal():
    # what k_resize's row cache (and the classify resize's row table: bit 31 = "y1 is the next row") relies on: y1 is y0 or y0 + 1, rows never go back
    rng = np.random.default_rng(3)
    sizes = [(1080, 764), (764, 540), (540, 382), (382, 270), (270, 191), (191, 135), (135, 95), (2160, 1527)]
    sizes += [(int(s), int(d)) for s, d in zip(rng.integers(2, 1500, 200), rng.integers(1, 1500, 200))]
    for sh, dh in sizes:
        scale_y = 1.0 / (dh / sh)
        prev = (-1, -1)
        for dy in range(dh):
            ya, yb = _row_table(dy, scale_y, sh)
            assert yb in (ya, ya + 1)
            assert ya >= prev[0] and yb >= prev[1]
            prev = (ya, yb)


# ---- k_nms: a chain's winner, all members at once ---------------------------------------------------------------------------------------------------------

def _winner_sequential(areas, T):
    """src/ER.cpp:464-497 as the old kernel path walks it: i over the chain, trail = chain[i], lead = chain[i + T]."""
    n = len(areas)
    if n < 1 + T:
        return None
    best, best_st, best_a = 0, 0.0, 0
    for i in range(n - T):
        a, bb = areas[i], areas[i + T]
        with np.errstate(divide="ignore"):
            st = np.float64(a) / np.float64(bb - a)         # 0 denominator -> +inf, as in the reference
        if i == 0 or st > best_st:
            best, best_st, best_a = i, st, a
        elif st == best_st and a < best_a:
            best, best_a = i, a
    return best


def _winner_parallel(areas, T):
    """The round-4 form: every member whose T-th ancestor is in the chain has a stability; max of its bit pattern, then the lowest member."""
    n = len(areas)
    cand = []
    for i in range(n - T):
        a, bb = areas[i], areas[i + T]
        with np.errstate(divide="ignore"):
            st = np.float64(a) / np.float64(bb - a)
        cand.append((int(np.float64(st).view(np.uint64)), i))
    if not cand:
        return None
    top = max(c[0] for c in cand)
    return min(i for bits, i in cand if bits == top)        # (levels rise along a chain: the lowest level is the smallest index)


@pytest.mark.parametrize("T", [0, 1, 2, 3, 5])
def test_chain_winner_in_parallel_is_the_sequential_one(T):
    rng = np.random.default_rng(100 + T)
    for trial in range(4000):
        n = int(rng.integers(1, 40))
        # box areas along a chain never shrink; plateaus (equal boxes -> 0 denominators, equal stabilities) are common
        steps = rng.choice([0, 0, 1, 2, 3, 10, 100, 5000], n)
        areas = (int(rng.integers(1, 50)) + np.cumsum(steps)).tolist()
        assert _winner_parallel(areas, T) == _winner_sequential(areas, T), (areas, T)
