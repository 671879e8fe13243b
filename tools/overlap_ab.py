"""Streaming turn through the host entry points (context stream of the highest priority + internal stream), device time between the
turn's two events (upload excluded, the flow copy excluded with zero_copy=0) for the "stream_overlap" settings 0 / 1 / 2, alternating.

    python3 tools/overlap_ab.py [--reps 40]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hackathonopticalflow_amd as ofa  # noqa: E402
from hackathonopticalflow_amd.synth import translated_pair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--levels", type=int, default=5)
    a = ap.parse_args()
    w, h = 1920, 1080
    fr = [translated_pair(h, w, 2001 + i)[0] for i in range(4)]
    eng = ofa.FarnebackEngine(w, h, 1, 0, levels=a.levels)
    out = ofa.pinned_empty((h, w, 2))
    eng.set_option("stream_zero_copy", 0)
    eng.stream_next(fr[0])
    for i in range(8):
        eng.stream_next(fr[(i + 1) % 4], out)
    for rnd in range(3):
        for ov in (0, 1, 2):
            eng.set_option("stream_overlap", ov)
            for i in range(4):
                eng.stream_next(fr[i % 4], out)
            ts, dev = [], []
            for i in range(a.reps):
                t0 = time.perf_counter()
                eng.stream_next(fr[(i + 2) % 4], out)
                ts.append((time.perf_counter() - t0) * 1e3)
                dev.append(eng.last_device_ms)
            print(f"round {rnd} stream_overlap={ov}: device median {np.median(dev) * 1e3:.1f} us (min {min(dev) * 1e3:.1f}), wall {np.median(ts):.4f} ms", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
