#!/usr/bin/env python3
"""Per-kernel hipEvent table of ONE streaming turn (one new 1080p frame): where the single-pair latency goes.

    python tools/streamprof.py [--w 1920 --h 1080 --levels 5 --reps 20]
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
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pair
    f0, f1, _ = translated_pair(a.h, a.w, 2001)
    f2, f3, _ = translated_pair(a.h, a.w, 2002)
    fr = [f0, f1, f2, f3]
    eng = ofa.FarnebackEngine(a.w, a.h, 1, 0, levels=a.levels)
    out = ofa.pinned_empty((a.h, a.w, 2))
    eng.stream_next(fr[0])
    for i in range(5):
        eng.stream_next(fr[(i + 1) % 4], out)
    for zc in (1, 0):
        eng.set_option("stream_zero_copy", zc)
        ts, dev = [], []
        for i in range(a.reps):
            t0 = time.perf_counter()
            eng.stream_next(fr[(i + 2) % 4], out)
            ts.append((time.perf_counter() - t0) * 1e3)
            dev.append(eng.last_device_ms)
        print(f"zero_copy={zc}: wall median {np.median(ts):.4f} ms (min {min(ts):.4f}), device {np.median(dev):.4f} ms")
    # pipelined submission: where does the host spend its time?
    outs = [ofa.pinned_empty((a.h, a.w, 2)) for _ in range(3)]
    pin = [ofa.pinned_empty((a.h, a.w), np.uint8) for _ in range(4)]
    for p_, f_ in zip(pin, fr):
        p_[...] = f_
    for name, src, pb in (("pageable frames", fr, 0), ("pinned frames", pin, 0), ("pinned frames, push kernel of 8 blocks", pin, 8),
                          ("pinned frames, push kernel of 32 blocks", pin, 32), ("pinned frames, push kernel of 256 blocks", pin, 256)):
        eng.set_option("push_blocks", pb)
        eng.stream_reset()
        eng.stream_submit(src[0], outs[0])
        for i in range(4):
            eng.stream_submit(src[(i + 1) % 4], outs[i % 3])
        eng.stream_wait(0)
        ts_sub, ts_wait = [], []
        t_all = time.perf_counter()
        for i in range(a.reps):
            t0 = time.perf_counter()
            eng.stream_submit(src[(i + 1) % 4], outs[i % 3])
            t1 = time.perf_counter()
            eng.stream_wait(1)
            t2 = time.perf_counter()
            ts_sub.append((t1 - t0) * 1e3)
            ts_wait.append((t2 - t1) * 1e3)
        eng.stream_wait(0)
        t_all = (time.perf_counter() - t_all) / a.reps * 1e3
        print(f"pipelined, {name}: {t_all:.4f} ms per frame; submit call {np.median(ts_sub):.4f} ms, wait(1) {np.median(ts_wait):.4f} ms")
    # two turns in flight
    outs4 = [ofa.pinned_empty((a.h, a.w, 2)) for _ in range(4)]
    eng.stream_reset()
    eng.stream_submit(pin[0], outs4[0])
    for i in range(6):
        eng.stream_submit(pin[(i + 1) % 4], outs4[i % 4])
        eng.stream_wait(2)
    t_all = time.perf_counter()
    for i in range(a.reps):
        eng.stream_submit(pin[(i + 1) % 4], outs4[i % 4])
        eng.stream_wait(2)
    eng.stream_wait(0)
    print(f"pipelined, pinned frames, TWO turns in flight: {(time.perf_counter() - t_all) / a.reps * 1e3:.4f} ms per frame")
    # the same loop through the Python class (what bench.py --config 2 --stream reports as pipelined_ms_per_frame)
    with ofa.FlowStream(levels=a.levels, pipelined=2) as stp:
        for i in range(8):
            stp.next(fr[i % 4])
        t_all = time.perf_counter()
        for i in range(a.reps):
            stp.next(fr[(i + 2) % 4])
        while stp.flush() is not None:
            pass
        print(f"pipelined, FlowStream(pipelined=2), pageable frames: {(time.perf_counter() - t_all) / a.reps * 1e3:.4f} ms per frame")
    with ofa.FlowStream(levels=a.levels, pipelined=True) as stp:
        for i in range(6):
            stp.next(fr[i % 4])
        t_all = time.perf_counter()
        for i in range(a.reps):
            stp.next(fr[(i + 2) % 4])
        stp.flush()
        print(f"pipelined, FlowStream(pipelined=True), pageable frames: {(time.perf_counter() - t_all) / a.reps * 1e3:.4f} ms per frame")
    eng.set_option("push_blocks", 0)
    eng.stream_reset()
    eng.stream_next(fr[0])
    eng.stream_next(fr[1], out)
    eng.profile_enable(True)
    for i in range(a.reps):
        eng.stream_next(fr[(i + 2) % 4], out)
    rows = eng.profile_read()
    tot = sum(r["ms"] for r in rows) / a.reps
    print(f"per-kernel events, copy path, mean of {a.reps} turns: sum {tot * 1e3:.1f} us in {sum(r['launches'] for r in rows) // a.reps} launches")
    for r in sorted(rows, key=lambda r: (-r["level"] if r["level"] < 31 else -99, r["stage"])):
        print(f"  {r['stage']:12s} L{r['level']:<2d} x{r['launches'] // a.reps}  {r['ms'] / a.reps * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
