#!/usr/bin/env python3
"""Offline randomized parity sweep (bigger than the seeded fuzz in tests/): random sizes, parameters and flags through the
whole dense pipeline on the GPU, compared bit for bit with the CPU oracle in the device summation order.

    python tests/bigfuzz_offline.py [seed] [cases]    (needs an MI355X; test infrastructure: imports oracle/ as the checker.
                                                       Not collected by pytest -- the seeded fuzz cases in test_gpu_parity.py are)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hackathonopticalflow_amd as H
from hackathonopticalflow_amd.synth import translated_pair, warped_pair
from oracle import oracle as O
O.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 123)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for i in range(N):
    w = int(rng.choice([rng.integers(33, 700), 64 * rng.integers(1, 12), 16 * rng.integers(3, 50)]))
    h = int(rng.choice([rng.integers(33, 500), 32 * rng.integers(2, 14), 16 * rng.integers(3, 30)]))
    kw = dict(levels=int(rng.integers(0, 6)), winsize=int(rng.integers(3, 27)), iterations=int(rng.integers(1, 4)),
              poly_n=int(rng.choice([3, 5, 5, 5, 7, 7])), poly_sigma=float(rng.choice([1.1, 1.2, 1.5])),
              pyr_scale=float(rng.choice([0.5, 0.5, 0.5, 0.6, 0.75, 0.8])), flags=int(rng.choice([0, 0, 0, 4, 256, 260])))
    os.environ["OFARN_DIRECT_MIN_FRAMES"] = str(int(rng.choice([0, 1000000])))
    if i % 3 == 0:      # a third of the cases on the FPV-like warped family (non-uniform sub-pixel flow + occluder)
        a, b, gt, _ = warped_pair(h, w, 50000 + i, zoom=float(rng.uniform(0.96, 1.08)), angle_deg=float(rng.uniform(-4, 4)),
                                  shift=(float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4))))
        tx, ty = (float(v) for v in gt[h // 2, w // 2])
    else:
        a, b, (tx, ty) = translated_pair(h, w, 50000 + i, max_shift=5)
    init = None
    if kw["flags"] & 4:
        init = (np.array([tx, ty], np.float32) + rng.standard_normal((h, w, 2)).astype(np.float32)).astype(np.float32)
    ref = O.farneback(a, b, box_mode=O.BOX_BLOCKED, init_flow=init, **kw)
    with H.FarnebackEngine(w, h, 1, **kw) as eng:
        got = eng.calc(a, b, None if init is None else init.copy())
        # the same pair as a streaming session (round 3): prime with a, then one turn with b -- must equal the pair call
        assert eng.stream_next(a, None if init is None else init.copy()) is None
        got_s = eng.stream_next(b, None if init is None else init.copy())
    if i % 4 == 1 and not (kw["flags"] & 4):
        # a small video-order batch through the wave scheduler (3 frames = 2 pairs, one wave or two), danger maps included:
        # pair 0 must equal the pair call, the maps must equal the reference's NumPy filter on the returned flow
        c = translated_pair(h, w, 70000 + i, max_shift=4)[0]
        gs = int(rng.choice([12, 20, 30]))
        with H.FarnebackEngine(w, h, int(rng.choice([1, 2])), grid_step=gs, **kw) as eng:
            fl, mk, vv = eng.calc_batch(np.stack([a, b, c]), H.PAIRS_CONSECUTIVE)
        P = len(H.grid_points(w, h, gs))
        okb = np.array_equal(fl[0], got)
        if P >= 2:
            for j in range(2):
                m_ref, v_ref = O.danger_map_numpy(fl[j], w, h, gs)
                okb = okb and np.array_equal(mk[j], m_ref) and np.array_equal(vv[j], v_ref)
        if not okb:
            bad += 1
            print("BATCH MISMATCH", w, h, kw, gs, flush=True)
    # round 4: the same pair through ofarn_calc_reuse (a miss = one pair turn, then a hit after the roles are swapped back) and, every
    # third case, in OpenCV's literal summation order ("box_order" = 1) against the oracle's OFO_BOX_RUNNING
    with H.FarnebackEngine(w, h, 1, **kw) as eng:
        g1 = eng.calc_reuse(a, b, None if init is None else init.copy())
        g2 = eng.calc_reuse(b, a, None if init is None else init.copy())        # prev = the held frame: a hit
        g3 = eng.calc_reuse(a, b, None if init is None else init.copy())        # again a hit
        hits, misses = eng.reuse_info()
        if not (np.array_equal(g1, got) and np.array_equal(g3, got) and (hits, misses) == (2, 1)):
            bad += 1
            print("REUSE MISMATCH", w, h, kw, hits, misses, flush=True)
        if i % 3 == 2:
            eng.set_option("box_order", 1)
            lit = eng.calc(a, b, None if init is None else init.copy())
            if not np.array_equal(lit, O.farneback(a, b, box_mode=O.BOX_RUNNING, init_flow=init, **kw)):
                bad += 1
                print("LITERAL-ORDER MISMATCH", w, h, kw, flush=True)
    if not np.array_equal(got_s, got):
        bad += 1
        print("STREAM MISMATCH", w, h, kw, float(np.abs(got_s - got).max()), flush=True)
    if not np.array_equal(got, ref):
        bad += 1
        print("MISMATCH", w, h, kw, os.environ["OFARN_DIRECT_MIN_FRAMES"], float(np.abs(got - ref).max()), flush=True)
    if (i + 1) % 100 == 0:
        print("...", i + 1, "cases,", bad, "mismatches", flush=True)
print("cases", N, "mismatches", bad)
