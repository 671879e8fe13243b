"""Soak run: many turns of every host-facing loop on one context set, watching device memory (hipMemGetInfo through torch) and the
process's resident set for growth.  Prints one line per phase; exits 1 if free device memory or RSS drifts by more than the limits.

    python3 tools/soak.py [--turns 3000] [--w 640 --h 480]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hackathonopticalflow_amd as H  # noqa: E402


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


def free_mb():
    return torch.cuda.mem_get_info()[0] / 2**20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--turns", type=int, default=3000)
    ap.add_argument("--w", type=int, default=640)
    ap.add_argument("--h", type=int, default=480)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (8, a.h, a.w), dtype=np.uint8)
    bgr = np.stack([frames, frames, frames], -1)
    torch.cuda.init()
    ok = True

    def phase(name, fn, turns):
        nonlocal ok
        fn(50)                                       # warm: lazily allocated buffers
        torch.cuda.synchronize()
        f0, r0, t0 = free_mb(), rss_mb(), time.perf_counter()
        fn(turns)
        torch.cuda.synchronize()
        df, dr = f0 - free_mb(), rss_mb() - r0
        bad = df > 8 or dr > 64
        ok &= not bad
        print(f"{name:34s} {turns:6d} turns  {1e3 * (time.perf_counter() - t0) / turns:7.3f} ms/turn  device +{df:.1f} MB  rss +{dr:.1f} MB"
              f"{'  <-- GROWTH' if bad else ''}", flush=True)

    with H.FlowStream(levels=3) as st:
        phase("FlowStream.next (gray)", lambda n: [st.next(frames[i & 7]) for i in range(n)], a.turns)
        phase("FlowStream.next (BGR)", lambda n: [st.next(bgr[i & 7]) for i in range(n)], a.turns)
        phase("FlowStream.next_view + view_lamps", lambda n: [(st.next_view(bgr[i & 7], rainbow=(i % 16 == 0)), st.view_lamps(over_frame=True))
                                                               for i in range(n)], a.turns)
        phase("FlowStream reset every 5 turns", lambda n: [(st.next(frames[i & 7]), st.reset() if i % 5 == 4 else None) for i in range(n)], a.turns)
    with H.FlowStream(levels=3, pipelined=True) as st:
        phase("FlowStream pipelined", lambda n: [st.next(frames[i & 7]) for i in range(n)] + [st.flush()], a.turns)
    phase("calculate_optical_flow (drop-in)", lambda n: [H.calculate_optical_flow(frames[i & 7], frames[(i + 1) & 7]) for i in range(n)], a.turns)
    with H.FarnebackEngine(a.w, a.h, 4, levels=3) as eng:
        phase("calc_batch (7 pairs, 2 waves)", lambda n: [eng.calc_batch(frames, H.PAIRS_CONSECUTIVE) for _ in range(n // 8 + 1)], a.turns)
        pts = H.grid_points(a.w, a.h, 30)
        phase("lk + danger_map + draw_lamps", lambda n: [(eng.lk(frames[0], frames[1], pts, winSize=(21, 21), maxLevel=2),
                                                          eng.draw_lamps(*eng.danger_map(np.zeros((a.h, a.w, 2), np.float32) + i), (a.h, a.w)))
                                                         for i in range(n // 4 + 1)], a.turns)
    phase("engine create / destroy", lambda n: [H.FarnebackEngine(a.w, a.h, 2, levels=3).close() for _ in range(n // 20 + 1)], a.turns)
    print("soak", "ok" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
