#!/usr/bin/env python3
"""tests/golden/interleave4_3x3.npz: a hand-derived known-answer vector for the vendor live view's 4-frame interleave
(opt_materials/software/XPR_Software.py:196-205 shift matrices M0..M3, :388-410 zero-insert + cv2.warpAffine + uint8 sum).

OpenCV is not installed in the build container, so this vector is pinned by the DOCUMENTED semantics, not by a reference run:
  * cv2.warpAffine(src, M, dsize) without WARP_INVERSE_MAP computes dst(x, y) = src(M^-1 (x, y)); for the pure translations
    M = [[1, 0, tx], [0, 1, ty]] that is dst(x, y) = src(x - tx, y - ty);
  * borderMode = BORDER_REFLECT_101 continues an axis of length n as  gfedcb|abcdefgh|gfedcba:  index -1 -> 1,  n -> n - 2;
  * np.sum(..., axis=2, dtype=np.uint8) adds the four planes modulo 256.
For 3x3 frames the HR planes are 6x6 with frame_k[i, j] at [2i, 2j] and zeros elsewhere.  Written out by hand:
  ty = 0 :  HR row y reads plane row      [0, 1, 2, 3, 4, 5]
  ty = +1:  HR row y reads plane row y-1: [-1 -> 1, 0, 1, 2, 3, 4]
  tx = 0 :  HR col x reads plane col      [0, 1, 2, 3, 4, 5]
  tx = -1:  HR col x reads plane col x+1: [1, 2, 3, 4, 5, 6 -> 4]
(M0..M3) = (tx, ty) = (0, 0), (0, +1), (-1, +1), (-1, 0).  The tables below are those four lines, nothing else is computed.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = {0: [0, 1, 2, 3, 4, 5], +1: [1, 0, 1, 2, 3, 4]}
COLS = {0: [0, 1, 2, 3, 4, 5], -1: [1, 2, 3, 4, 5, 4]}
M = [(0, 0), (0, +1), (-1, +1), (-1, 0)]  # (tx, ty)

frames = np.array([[[10, 20, 30], [40, 50, 60], [70, 80, 90]],
                   [[1, 2, 3], [4, 5, 6], [7, 8, 9]],
                   [[200, 210, 220], [230, 240, 250], [255, 128, 64]],
                   [[100, 101, 102], [103, 104, 105], [106, 107, 108]]], dtype=np.uint8)
out = np.zeros((6, 6), dtype=np.uint32)
for k, (tx, ty) in enumerate(M):
    plane = np.zeros((6, 6), dtype=np.uint32)
    for i in range(3):
        for j in range(3):
            plane[2 * i, 2 * j] = frames[k, i, j]
    for y in range(6):
        for x in range(6):
            out[y, x] += plane[ROWS[ty][y], COLS[tx][x]]
expected = (out % 256).astype(np.uint8)
dst = os.path.join(ROOT, "tests", "golden", "interleave4_3x3.npz")
np.savez(dst, frames=frames, expected=expected)
print(expected)
