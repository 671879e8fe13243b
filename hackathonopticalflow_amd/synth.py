"""Synthetic frame pairs.

The reference's demo videos are H.264 and cannot be decoded in the build or GPU containers, so
every test and benchmark runs on these generators, both on Gaussian-low-passed white noise quantised
to uint8:

* ``translated_pair``: ``next`` is an integer translation of ``prev`` (ground-truth flow is exactly (tx, ty));
* ``warped_pair``: the reference's actual use case, FPV forward flight -- ``next`` is ``prev`` zoomed about a
  focus of expansion, rotated a little and shifted by a sub-pixel amount, with an independently moving
  occluding patch on top (pathfinder_viewer.py:162-168 equalises moduli by distance to the centre for exactly
  this radial field).  The flow is non-uniform, sub-pixel and discontinuous at the patch: the bilinear gather of
  FarnebackUpdateMatrices is then neither aligned nor coalesced, and its out-of-image branch fires along whole
  borders instead of a few columns.
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


def warped_pair(height, width, seed, zoom=1.03, angle_deg=0.8, shift=(1.6, -0.7), focus=None, occluder=True, sigma=4.0):
    """FPV-like pair.  Background motion: q = c + zoom * Rot(angle) (p - c) + shift about the focus c (default: the
    reference's own centre, int(W/2), int(H/2), pathfinder_viewer.py:252-253); occluder: an elliptical patch of a
    second texture moving by its own integer vector.

    Returns (prev u8[H,W], next u8[H,W], flow float32[H,W,2] ground truth at prev's pixels,
             valid bool[H,W]: background pixels visible in both frames and at least 8 px from the patch)."""
    from scipy.ndimage import binary_dilation, map_coordinates
    rng = np.random.default_rng(seed)
    cx, cy = (float(int(width / 2)), float(int(height / 2))) if focus is None else (float(focus[0]), float(focus[1]))
    th = np.deg2rad(angle_deg)
    A = zoom * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    t = np.asarray(shift, np.float64)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    # ground truth at prev's pixels
    dxp, dyp = xs - cx, ys - cy
    fx = A[0, 0] * dxp + A[0, 1] * dyp + t[0] - dxp
    fy = A[1, 0] * dxp + A[1, 1] * dyp + t[1] - dyp
    pad = int(np.ceil(max(np.abs(fx).max(), np.abs(fy).max()))) + 12
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.standard_normal((height + 2 * pad, width + 2 * pad)), sigma, mode="wrap")
    lo, hi = float(base.min()), float(base.max())
    base = (base - lo) * (255.0 / (hi - lo))
    prev_f = base[pad:pad + height, pad:pad + width]
    # next(q) = base(p) with p = c + A^-1 (q - c - t)
    Ai = np.linalg.inv(A)
    qx, qy = xs - cx - t[0], ys - cy - t[1]
    px = cx + Ai[0, 0] * qx + Ai[0, 1] * qy
    py = cy + Ai[1, 0] * qx + Ai[1, 1] * qy
    next_f = map_coordinates(base, [py + pad, px + pad], order=3, mode="reflect")
    flow = np.stack([fx, fy], -1).astype(np.float32)
    valid = np.ones((height, width), bool)
    if occluder:
        tex = gaussian_filter(rng.standard_normal((height, width)), sigma * 0.75, mode="wrap")
        tex = (tex - tex.min()) * (255.0 / (tex.max() - tex.min()))
        ox, oy = 0.62 * width, 0.42 * height
        rx, ry = max(4.0, 0.09 * width), max(4.0, 0.12 * height)
        d = np.array([int(rng.integers(3, 8)), -int(rng.integers(2, 6))])
        in_prev = ((xs - ox) / rx) ** 2 + ((ys - oy) / ry) ** 2 <= 1.0
        in_next = ((xs - ox - d[0]) / rx) ** 2 + ((ys - oy - d[1]) / ry) ** 2 <= 1.0
        prev_f = np.where(in_prev, tex, prev_f)
        moved = np.roll(tex, (int(d[1]), int(d[0])), axis=(0, 1))
        next_f = np.where(in_next, moved, next_f)
        flow[in_prev] = d.astype(np.float32)
        valid &= ~binary_dilation(in_prev | in_next, iterations=8)
    prev = np.clip(np.round(prev_f), 0, 255).astype(np.uint8)
    nxt = np.clip(np.round(next_f), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(prev), np.ascontiguousarray(nxt), flow, valid


def warped_pairs(n_pairs, height, width, seed0, **kw):
    """frames u8[2*n_pairs,H,W] (prev0, next0, ...) of warped_pair with zoom / rotation / shift varied per pair."""
    frames = np.empty((2 * n_pairs, height, width), np.uint8)
    flows = np.empty((n_pairs, height, width, 2), np.float32)
    for i in range(n_pairs):
        r = np.random.default_rng(seed0 + 7919 * i)
        p, n, f, _ = warped_pair(height, width, seed0 + i, zoom=float(r.uniform(1.005, 1.04)),
                                 angle_deg=float(r.uniform(-1.5, 1.5)),
                                 shift=(float(r.uniform(-3, 3)), float(r.uniform(-3, 3))), **kw)
        frames[2 * i], frames[2 * i + 1], flows[i] = p, n, f
    return frames, flows
