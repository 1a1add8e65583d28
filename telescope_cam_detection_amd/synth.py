"""Synthetic frames for benchmarks and parity tests (there is no camera or dataset offline).

`noise_frame` is the reference's own benchmark input (tests/test_inference.py:76 of the
reference: np.random.randint(0, 255, (H, W, 3), uint8)); `scene_frame` adds structure
(gradients, rectangles, discs) so feature maps are not spatially uniform and the two top-k
selections are not decided by rounding noise alone.
"""
from __future__ import annotations

import numpy as np


def noise_frame(seed: int, h: int, w: int) -> np.ndarray:
    """uniform uint8 noise, upper-exclusive 255, HWC BGR."""
    return np.random.default_rng(seed).integers(0, 255, (h, w, 3), dtype=np.uint8)


def scene_frame(seed: int, h: int, w: int, n_objects: int = 14) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, 3), np.float32)
    for c in range(3):
        a, b, c0 = rng.uniform(-0.15, 0.15, 2).tolist() + [rng.uniform(60, 180)]
        img[:, :, c] = c0 + a * xx * (255.0 / w) + b * yy * (255.0 / h)
    for _ in range(n_objects):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sx, sy = rng.uniform(0.03, 0.25) * w, rng.uniform(0.03, 0.25) * h
        col = rng.uniform(0, 255, 3).astype(np.float32)
        if rng.random() < 0.5:
            m = (np.abs(xx - cx) < sx) & (np.abs(yy - cy) < sy)
        else:
            m = ((xx - cx) / sx) ** 2 + ((yy - cy) / sy) ** 2 < 1.0
        img[m] = col
    img += rng.normal(0, 6.0, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_frame(kind: str, seed: int, h: int, w: int) -> np.ndarray:
    return noise_frame(seed, h, w) if kind == "noise" else scene_frame(seed, h, w)
