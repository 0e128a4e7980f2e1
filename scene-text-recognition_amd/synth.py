"""Deterministic synthetic frames for tests and bench.py.

All randomness comes from splitmix64 used as a counter-based hash, so a frame is a pure
function of (kind, seed, width, height) and can be regenerated anywhere with numpy alone:

  S-text  : smooth ramp 96 + 64x/W + 32y/H with +-4 noise, plus W*H/2500 glyph-like hollow
            boxes with a centre bar (dark 20-59 or bright 200-239, +-3 noise), drawn with
            independent per-colour offsets.  Never reaches the sentinel level (p >= 252).
  S-noise : iid uniform 0..255 (stress; hits the level-32 sentinel of SURVEY A.2 constantly).
  S-flat  : constant 128.
  S-ties  : S-text plus, in every eighth frame (`every`), one "double L" glyph: a grey box holding two dark L-shaped strokes (left + bottom,
            top + right) that do not touch, each with a core one level below its rim.  Both strokes' boxes cover more than 0.7 of
            the grey box, so both child chains pass the overlap test of non_maximum_supression (src/ER.cpp:455-462) on the same
            parent: an NMS sibling tie whose two outcomes give DIFFERENT pools -- the case the reference decides by its flood
            order.  The Y planes of the first two or three pyramid levels of such a frame need the order walked: with every = 8 that is
            1.2 % of the planes of a pyr3x8 batch, with every = 3 3.1 %.
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output step applied element-wise to uint64 states."""
    with np.errstate(over="ignore"):
        z = (x + _GOLD).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _hash(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx.astype(np.uint64) * _GOLD)


def frame_seed(frame_index: int) -> int:
    return 0x5EED0000 + int(frame_index)


def snoise_bgr(seed: int, w: int, h: int) -> np.ndarray:
    idx = np.arange(w * h * 3, dtype=np.uint64)
    return (_hash(seed, idx) >> np.uint64(56)).astype(np.uint8).reshape(h, w, 3)


def sflat_bgr(seed: int, w: int, h: int, value: int = 128) -> np.ndarray:
    return np.full((h, w, 3), value, np.uint8)


def stext_bgr(seed: int, w: int, h: int) -> np.ndarray:
    n = w * h
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = (96 + (64 * xx) // max(w, 1) + (32 * yy) // max(h, 1)).astype(np.int32)
    img = np.empty((h, w, 3), np.int32)
    for c in range(3):
        nz = (_hash(seed, np.arange(n, dtype=np.uint64) + np.uint64(c * n)) % np.uint64(9)).astype(np.int32) - 4
        img[:, :, c] = ramp + nz.reshape(h, w)
    n_glyph = n // 2500
    g = _hash(seed ^ 0xA5A5A5A5, np.arange(n_glyph * 12, dtype=np.uint64)).reshape(n_glyph, 12)
    for i in range(n_glyph):
        r = [int(v) for v in g[i]]
        gw = 6 + r[0] % 60
        gh = 10 + r[1] % 80
        if gw >= w or gh >= h:
            continue
        st = 2 + r[2] % 6
        st = max(1, min(st, gw // 3, gh // 3))
        x0 = r[3] % (w - gw)
        y0 = r[4] % (h - gh)
        base = (20 + r[5] % 40) if (r[6] & 1) else (200 + r[5] % 40)
        mask = np.zeros((gh, gw), bool)
        mask[:st, :] = True
        mask[-st:, :] = True
        mask[:, :st] = True
        mask[:, -st:] = True
        mid = gh // 2
        mask[mid - st // 2: mid - st // 2 + st, :] = True
        pn = (_hash(seed ^ (0x1234 + i), np.arange(gw * gh * 3, dtype=np.uint64)) % np.uint64(7)).astype(np.int32) - 3
        pn = pn.reshape(gh, gw, 3)
        for c in range(3):
            off = (r[7 + c] % 17) - 8
            val = np.clip(base + off + pn[:, :, c], 0, 251)
            sub = img[y0:y0 + gh, x0:x0 + gw, c]
            sub[mask] = val[mask]
    return np.clip(img, 0, 251).astype(np.uint8)


def _erode4(m: np.ndarray) -> np.ndarray:
    e = m.copy()
    e[1:, :] &= m[:-1, :]; e[:-1, :] &= m[1:, :]; e[:, 1:] &= m[:, :-1]; e[:, :-1] &= m[:, 1:]
    e[0, :] = e[-1, :] = False; e[:, 0] = e[:, -1] = False
    return e


def draw_tie_glyph(img: np.ndarray, x0: int, y0: int, gw: int, gh: int, st1: int, st2: int, fill: int, box: int = 80, moat: int = 150) -> None:
    """The double-L glyph of S-ties into img[..., :] (all colour channels alike) with its top-left corner at (x0, y0)."""
    h, w = img.shape[:2]
    img[max(0, y0 - 3):min(h, y0 + gh + 3), max(0, x0 - 3):min(w, x0 + gw + 3)] = moat      # keeps neighbouring glyphs out of the box's component
    sub = img[y0:y0 + gh, x0:x0 + gw]
    sub[...] = box
    gap, m = 2, 1
    l1 = np.zeros((gh, gw), bool)
    l2 = np.zeros((gh, gw), bool)
    l1[m:gh - m, m:m + st1] = True
    l1[gh - m - st1:gh - m, m:gw - m - gap - st2] = True                 # left + bottom
    l2[m:m + st2, m + st1 + gap:gw - m] = True
    l2[m:gh - m - st1 - gap, gw - m - st2:gw - m] = True                 # top + right
    for l in (l1, l2):
        sub[l] = fill + 10
        sub[_erode4(l)] = fill


def sties_bgr(seed: int, w: int, h: int, every: int = 8) -> np.ndarray:
    img = stext_bgr(seed, w, h)
    if seed % every == 0 and w >= 160 and h >= 160:
        r = [int(v) for v in _hash(seed ^ 0x71E5, np.arange(6, dtype=np.uint64))]
        gw, gh = 76 + r[0] % 36, 76 + r[1] % 36
        draw_tie_glyph(img, r[2] % (w - gw), r[3] % (h - gh), gw, gh, 3 + r[4] % 3, 3 + (r[4] >> 8) % 3, 20 + r[5] % 30)
    return img


KINDS = {"text": stext_bgr, "noise": snoise_bgr, "flat": sflat_bgr, "ties": sties_bgr}


def frames_bgr(kind: str, first_frame: int, n_frames: int, w: int, h: int) -> np.ndarray:
    """(n_frames, h, w, 3) uint8, frame i uses seed 0x5EED0000 + first_frame + i."""
    fn = KINDS[kind]
    return np.stack([fn(frame_seed(first_frame + i), w, h) for i in range(n_frames)])


def nv12_from_bgr(bgr: np.ndarray) -> np.ndarray:
    """What a decoder would hand over for this frame: (h * 3 // 2, w) uint8 -- the luma plane, then interleaved Cb/Cr at half
    resolution.  Luma / chroma with the integer BGR->YCrCb formula of the path (OpenCV 8-bit, shift 14), chroma averaged over its
    2 x 2 block ((a+b+c+d+2) >> 2).  Input data for the NV12 entry point only; w and h even."""
    h, w = bgr.shape[:2]
    assert w % 2 == 0 and h % 2 == 0
    b, g, r = (bgr[..., k].astype(np.int32) for k in range(3))
    y = np.clip((1868 * b + 9617 * g + 4899 * r + 8192) >> 14, 0, 255)
    cr = np.clip(((r - y) * 11682 + (128 << 14) + 8192) >> 14, 0, 255)
    cb = np.clip(((b - y) * 9241 + (128 << 14) + 8192) >> 14, 0, 255)
    box = lambda p: (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + 2) >> 2
    out = np.empty((h + h // 2, w), np.uint8)
    out[:h] = y
    out[h:, 0::2] = box(cb)
    out[h:, 1::2] = box(cr)
    return out


def gray(bgr: np.ndarray) -> np.ndarray:
    """Cheap deterministic grey plane for single-plane tests: the G channel."""
    return np.ascontiguousarray(bgr[..., 1])
