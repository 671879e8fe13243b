"""The reference's frame loop through the drop-in, literally (DenseOF.py:519-525):

    flow = calculate_optical_flow(prev_gray, gray);  prev_gray = gray

timed per frame at 1920x1080 (levels 5 as BASELINE config 2, and the function's own default levels 3), next to FlowStream.next.

    python3 tools/dropin_loop.py [--frames 120] [--w 1920 --h 1080]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hackathonopticalflow_amd as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    a = ap.parse_args()
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (a.h + 64, a.w + 64), dtype=np.uint8)
    frames = [np.ascontiguousarray(base[i % 32:i % 32 + a.h, (3 * i) % 32:(3 * i) % 32 + a.w]) for i in range(16)]
    for levels in (5, 3):
        def loop(n, keep=False):
            prev, kept, s = frames[0], [], 0.0
            for i in range(1, n + 1):
                gray = frames[i & 15].copy()          # a NEW array per frame, as cap.read() + cvtColor give
                t0 = time.perf_counter()
                flow = H.calculate_optical_flow(prev, gray, levels=levels)
                s += time.perf_counter() - t0
                prev = gray
                if keep:
                    kept.append(flow)
            return 1e3 * s / n
        loop(20)
        ms = loop(a.frames)
        with H.FlowStream(levels=levels) as st:
            st.next(frames[0])
            for i in range(1, 20):
                st.next(frames[i & 15])
            t0 = time.perf_counter()
            for i in range(a.frames):
                st.next(frames[i & 15])
            ms_s = 1e3 * (time.perf_counter() - t0) / a.frames
        ms_keep = loop(24, keep=True)
        # the miss path: the caller edits the held frame in place before every call (one byte), so every call starts over
        def loop_edited(n):
            prev, s = frames[0].copy(), 0.0
            for i in range(1, n + 1):
                gray = frames[i & 15].copy()
                prev[(7 * i) % a.h, (13 * i) % a.w] ^= 1
                t0 = time.perf_counter()
                H.calculate_optical_flow(prev, gray, levels=levels)
                s += time.perf_counter() - t0
                prev = gray
            return 1e3 * s / n
        ms_edit = loop_edited(a.frames)
        # unrelated pairs (no loop at all): every call is a miss that the sampled look catches before anything is launched
        def loop_unrelated(n):
            s = 0.0
            for i in range(1, n + 1):
                p_, g_ = frames[(5 * i) & 15].copy(), frames[(5 * i + 3) & 15].copy()
                t0 = time.perf_counter()
                H.calculate_optical_flow(p_, g_, levels=levels)
                s += time.perf_counter() - t0
            return 1e3 * s / n
        ms_unrel = loop_unrelated(a.frames)
        from hackathonopticalflow_amd import ofarn
        hm = [sl.eng.reuse_info() for e in ofarn._engines.values() for sl in e.slots if sl.eng is not None]
        mode = "exact byte compare" if ofarn._DROPIN_REUSE else "OFARN_DROPIN_REUSE=0 (no frame held)"
        print(f"{a.w}x{a.h} levels={levels} [{mode}]: dropin_loop_ms_per_frame {ms:.3f}  (caller keeps every flow: {ms_keep:.3f}; "
              f"held frame edited in place (one byte) before every call: {ms_edit:.3f}; unrelated pairs: {ms_unrel:.3f})   FlowStream.next {ms_s:.3f} ms/frame   "
              f"reuse (hits, misses) per context: {hm}", flush=True)


if __name__ == "__main__":
    main()
