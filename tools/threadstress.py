"""Thread stress of the Python mirror: several threads at once, each running the reference's loop through the drop-in
(calculate_optical_flow(prev, gray); prev = gray), a FlowStream of its own and single pair calls, on frame sizes that partly coincide
(threads with the same size share the drop-in's cached contexts: two per key, a third thread waits).  Every result is compared with the
one a single thread computed beforehand.

    python3 tools/threadstress.py [--threads 6] [--turns 200]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hackathonopticalflow_amd as H  # noqa: E402
from hackathonopticalflow_amd.synth import translated_pair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--turns", type=int, default=200)
    a = ap.parse_args()
    sizes = [(320, 240), (320, 240), (320, 240), (640, 480), (416, 234), (640, 480), (320, 240), (1280, 720)]
    vids, refs = [], []
    for t in range(a.threads):
        w, h = sizes[t % len(sizes)]
        frames = [translated_pair(h, w, 500 + 7 * t + i, max_shift=3)[0] for i in range(6)]
        vids.append(frames)
        with H.FarnebackEngine(w, h, 1) as eng:
            refs.append([eng.calc(frames[i], frames[(i + 1) % 6]) for i in range(6)])
    errors, counts = [], [0] * a.threads
    start = threading.Barrier(a.threads)

    def worker(t):
        frames, ref = vids[t], refs[t]
        try:
            start.wait()
            with H.FlowStream() as st:
                prev = frames[0]
                st.next(frames[0])
                for k in range(1, a.turns + 1):
                    i = k % 6
                    gray = frames[i].copy()
                    flow = H.calculate_optical_flow(prev, gray)
                    if not np.array_equal(flow, ref[(i - 1) % 6]):
                        raise AssertionError(f"thread {t} turn {k}: drop-in result differs")
                    prev = gray
                    f2 = st.next(frames[i])
                    if not np.array_equal(f2, ref[(i - 1) % 6]):
                        raise AssertionError(f"thread {t} turn {k}: FlowStream result differs")
                    if k % 17 == 0:
                        mask, v = H.danger_map(flow)
                        layer = H.draw_sparse_lamps(mask, v, flow.shape[:2])
                        assert layer.shape == flow.shape[:2] + (3,)
                    counts[t] = k
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(a.threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    print(f"{a.threads} threads x {a.turns} turns (2 flow calls each): {dt:.1f} s, turns done {counts}, errors: {errors if errors else 'none'}")
    return 1 if errors or min(counts) < a.turns else 0


if __name__ == "__main__":
    sys.exit(main())
