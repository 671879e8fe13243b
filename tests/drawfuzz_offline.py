#!/usr/bin/env python3
"""Offline randomized sweep of the drawing entry points (draw_flow image, get_flow_lk layer, obstacle layer, cv2.add) against the
restated cv2 rasters: random frame sizes, grid / arrow steps, radii and flow magnitudes (arrows that cross, leave the image, have
zero length; discs clipped at the border).

    python tests/drawfuzz_offline.py [seed] [cases]     (needs an MI355X; test infrastructure: imports oracle/ as the checker)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hackathonopticalflow_amd as H
from oracle import oracle as O

O.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for i in range(N):
    w, h = int(rng.integers(8, 400)), int(rng.integers(8, 300))
    scale = float(rng.choice([0.5, 3, 12, 60, 500]))
    flow = (rng.standard_normal((h, w, 2)) * scale).astype(np.float32)
    if i % 5 == 0:
        flow[rng.integers(0, h), rng.integers(0, w)] = (1e6, -1e6)
    step = int(rng.integers(1, 40))
    ok = np.array_equal(H.draw_flow((h, w), flow, step), O.draw_flow_numpy((h, w), flow, step))
    gs = int(rng.integers(3, 45))
    radius = int(rng.integers(0, min(31, (gs - 1) // 2) + 1))
    pts = O.grid_points_numpy(w, h, gs)
    with H.FarnebackEngine(w, h, 1, grid_step=gs) as eng:
        base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if len(pts) >= 2:
            vec = (rng.standard_normal((len(pts), 2)) * scale).astype(np.float32)
            mask, _mod, iflow, ipts = O.vector_filter_numpy(vec, pts, w, h, 0)
            m2, v2, if2 = eng.vector_filter(vec, w, h, return_flow=True)
            ok = ok and np.array_equal(m2.astype(bool), mask) and np.array_equal(if2, iflow)
            dbf = bool(rng.integers(0, 2))
            layer = eng.draw_vectors(if2, m2, (h, w), dbf)
            ok = ok and np.array_equal(layer, O.get_flow_lk_layer_numpy(mask, iflow, ipts, w, h, dbf))
            ok = ok and np.array_equal(eng.add_u8(base, layer), O.cv_add_u8(base, layer))
            lamps = eng.draw_lamps(m2, v2, (h, w), radius=radius, base=base)
            ref = O.cv_add_u8(base, O.draw_sparse_lamps_numpy(iflow[mask], ipts[mask], w, h, radius))
            ok = ok and np.array_equal(lamps, ref)
    if not ok:
        bad += 1
        print("MISMATCH", i, w, h, step, gs, radius, scale, flush=True)
    if (i + 1) % 100 == 0:
        print("...", i + 1, "cases,", bad, "mismatches", flush=True)
print("cases", N, "mismatches", bad)
