"""Synthetic "translated smooth noise" frame pairs.

The reference's demo videos are H.264 and cannot be decoded in the build or GPU containers, so
every test and benchmark runs on this generator: Gaussian-low-passed white noise, quantised to
uint8, with ``next`` an integer translation of ``prev`` (ground-truth flow is exactly (tx, ty)).
"""
from __future__ import annotations

import numpy as np

PAD = 16


def _smooth_noise(rng, h, w, sigma=4.0):
    from scipy.ndimage import gaussian_filter
    base = rng.standard_normal((h, w)).astype(np.float32)
    base = gaussian_filter(base, sigma, mode="wrap")
    lo, hi = float(base.min()), float(base.max())
    return np.round((base - lo) * (255.0 / (hi - lo))).astype(np.uint8)


def translated_pair(height, width, seed, max_shift=8, sigma=4.0):
    """Returns (prev u8[H,W], next u8[H,W], (tx, ty))."""
    rng = np.random.default_rng(seed)
    base = _smooth_noise(rng, height + 2 * PAD, width + 2 * PAD, sigma)
    tx, ty = (int(v) for v in rng.integers(-max_shift, max_shift + 1, size=2))
    prev = base[PAD:PAD + height, PAD:PAD + width]
    nxt = base[PAD - ty:PAD - ty + height, PAD - tx:PAD - tx + width]
    return np.ascontiguousarray(prev), np.ascontiguousarray(nxt), (tx, ty)


def translated_pairs(n_pairs, height, width, seed0, max_shift=8, unique=None):
    """frames u8[2*n_pairs,H,W] laid out (prev0, next0, prev1, next1, ...) and shifts int[n_pairs,2].

    ``unique`` < n_pairs generates that many distinct pairs and tiles them (run time of the flow
    pipeline is data independent; stated wherever it is used)."""
    unique = n_pairs if unique is None else min(unique, n_pairs)
    frames = np.empty((2 * n_pairs, height, width), np.uint8)
    shifts = np.empty((n_pairs, 2), np.int64)
    for i in range(unique):
        p, n, s = translated_pair(height, width, seed0 + i, max_shift)
        frames[2 * i], frames[2 * i + 1], shifts[i] = p, n, s
    for i in range(unique, n_pairs):
        j = i % unique
        frames[2 * i], frames[2 * i + 1], shifts[i] = frames[2 * j], frames[2 * j + 1], shifts[j]
    return frames, shifts
