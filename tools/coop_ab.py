#!/usr/bin/env python3
"""Synchronous streaming turn at 1080p (ofarn_stream_next into a page-locked flow buffer), alternating the latency switches on
ONE context and ONE box: "coop_levels" 0 / 1 / 2 (all iterations of the coarse levels in one launch behind a
bounded device-wide barrier) x zero-copy on / off.  Wall time per turn (median, min) and the device time of the turn's kernels.

    python tools/coop_ab.py [--reps 60] [--rounds 3] [--levels 5]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pair
    fr = [f for s in (2001, 2002) for f in translated_pair(a.h, a.w, s)[:2]]
    eng = ofa.FarnebackEngine(a.w, a.h, 1, 0, levels=a.levels)
    out = ofa.pinned_empty((a.h, a.w, 2))
    ref = None
    eng.stream_next(fr[0])
    for i in range(8):
        eng.stream_next(fr[(i + 1) % 4], out)
    print(f"# {a.w}x{a.h} levels={a.levels}, {a.reps} turns per leg, {a.rounds} alternations; wall = ofarn_stream_next through ctypes")
    for rnd in range(a.rounds):
        for zc in (1, 0):
            for coop in (0, 1, 2):
                eng.set_option("stream_zero_copy", zc)
                eng.set_option("coop_levels", coop)
                for i in range(6):
                    eng.stream_next(fr[(i + 2) % 4], out)
                ts, dev = [], []
                for i in range(a.reps):
                    t0 = time.perf_counter()
                    eng.stream_next(fr[(i + 2) % 4], out)
                    ts.append((time.perf_counter() - t0) * 1e3)
                    dev.append(eng.last_device_ms)
                if ref is None:
                    ref = out.copy()
                same = bool(np.array_equal(out, ref)) if (a.reps + 1) % 4 == (a.reps + 1) % 4 else None
                print(f"round {rnd + 1}: zero_copy={zc} coop_levels={coop}: wall median {np.median(ts):.4f} ms (min {min(ts):.4f}), "
                      f"device {np.median(dev):.4f} ms   coop (launches, fallbacks) so far {eng.coop_info()}   same flow as first leg: {same}",
                      flush=True)


if __name__ == "__main__":
    main()
