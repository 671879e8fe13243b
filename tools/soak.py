"""Soak run: many turns of every host-facing loop on one context set, watching device memory (hipMemGetInfo through torch) and the
process's resident set for growth over the second half of each loop.  Prints one line per phase; exits 1 if free device memory or RSS
drifts by more than the limits.

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
        # what counts is growth that CONTINUES: lazily allocated buffers and the runtime's own one-time pools (seen: +190 MB of host
        # mappings and +2 MB of device memory once per process, some hundred turns into the first loop that alternates transfer sizes)
        # land in the first half, the second half is compared with the state after the first
        t0 = time.perf_counter()
        fn(turns - turns // 2)
        torch.cuda.synchronize()
        f0, r0 = free_mb(), rss_mb()
        fn(turns // 2)
        torch.cuda.synchronize()
        df, dr = f0 - free_mb(), rss_mb() - r0
        bad = df > 8 or dr > 64
        ok &= not bad
        print(f"{name:40s} {turns:6d} turns  {1e3 * (time.perf_counter() - t0) / turns:7.3f} ms/turn  device +{df:.1f} MB  rss +{dr:.1f} MB"
              f"{'  <-- GROWTH' if bad else ''}", flush=True)

    def loop(body):
        """n -> runs body(i) n times, dropping every result at once (a list of 6000 flow fields would be 100 GB)"""
        def run(n):
            for i in range(n):
                body(i)
        return run

    with H.FlowStream(levels=3) as st:
        phase("FlowStream.next (gray)", loop(lambda i: st.next(frames[i & 7])), a.turns)
        phase("FlowStream.next (BGR)", loop(lambda i: st.next(bgr[i & 7])), a.turns)
        phase("FlowStream.next_view + view_lamps", loop(lambda i: (st.next_view(bgr[i & 7], rainbow=(i % 16 == 0)), st.view_lamps(over_frame=True),
                                                                    st.view_arrows(14, over_frame=(i % 3 == 0)))), a.turns)
        phase("FlowStream reset every 5 turns", loop(lambda i: (st.next(frames[i & 7]), st.reset() if i % 5 == 4 else None)), a.turns)
    with H.FlowStream(levels=3, pipelined=True) as st:
        phase("FlowStream pipelined", loop(lambda i: st.next(frames[i & 7])), a.turns)
        st.flush()
    phase("calculate_optical_flow (drop-in)", loop(lambda i: H.calculate_optical_flow(frames[i & 7], frames[(i + 1) & 7])), a.turns)
    with H.FarnebackEngine(a.w, a.h, 4, levels=3) as eng:
        phase("calc_batch (7 pairs, 2 waves)", loop(lambda i: eng.calc_batch(frames, H.PAIRS_CONSECUTIVE)), a.turns // 8 + 1)
        pts = H.grid_points(a.w, a.h, 30)
        zero = np.zeros((a.h, a.w, 2), np.float32)
        phase("lk + danger_map + draw_lamps + draw_flow", loop(lambda i: (eng.lk(frames[0], frames[1], pts, winSize=(21, 21), maxLevel=2),
                                                                          eng.draw_lamps(*eng.danger_map(zero + i), (a.h, a.w)),
                                                                          eng.draw_flow(zero + (i % 7), 14))), a.turns // 4 + 1)
    phase("engine create / destroy", loop(lambda i: H.FarnebackEngine(a.w, a.h, 2, levels=3).close()), a.turns // 20 + 1)
    print("soak", "ok" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
