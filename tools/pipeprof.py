#!/usr/bin/env python3
"""The pipelined frame loop alone (ofarn_stream_submit / ofarn_stream_wait, 1080p), for a timeline trace:

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -o run -- python3 tools/pipeprof.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pair
    w, h = 1920, 1080
    f0, f1, _ = translated_pair(h, w, 2001)
    fr = [f0, f1]
    eng = ofa.FarnebackEngine(w, h, 1, 0, levels=5)
    outs = [ofa.pinned_empty((h, w, 2)) for _ in range(3)]
    eng.stream_submit(fr[0], outs[0])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    t0 = time.perf_counter()
    for i in range(n):
        eng.stream_submit(fr[(i + 1) % 2], outs[i % 3])
        eng.stream_wait(1)
    eng.stream_wait(0)
    print(f"{(time.perf_counter() - t0) / n * 1e3:.4f} ms per frame")
    eng.close()


if __name__ == "__main__":
    main()
