#!/usr/bin/env python3
"""Mint tests/golden/icdar_crops.npz: a few small BGR crops of the reference's sample photographs.

Development container only (needs /root/reference and Pillow).  The reference ships res/ICDAR2015_test/*.jpg as
sample inputs (data, not code); natural-image statistics exercise the kernels differently from the synthetic
frames (long smooth gradients, JPEG block noise, real glyphs).  Crops are stored as raw uint8 BGR arrays.
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/res/ICDAR2015_test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icdar_crops.npz")
PICKS = [("img_1.jpg", 300, 200, 320, 240), ("img_12.jpg", 100, 100, 320, 240), ("img_108.jpg", 0, 0, 256, 192),
         ("img_45.jpg", 200, 150, 384, 160)]


def main():
    out = {}
    names = sorted(os.listdir(SRC))
    for k, (name, x, y, w, h) in enumerate(PICKS):
        if name not in names:
            name = names[(k * 37) % len(names)]
        im = np.asarray(Image.open(os.path.join(SRC, name)).convert("RGB"))
        H, W, _ = im.shape
        x, y = min(x, max(0, W - w)), min(y, max(0, H - h))
        crop = im[y:y + h, x:x + w, ::-1]           # RGB -> BGR, the order cv::imread gives (src/utils.cpp:31)
        out[f"crop{k}"] = np.ascontiguousarray(crop)
        print(name, crop.shape)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
