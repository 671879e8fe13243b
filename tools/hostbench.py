#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer batch entry point (ofarn_calc_batch): NumPy frames in, NumPy flow +
danger maps out, with pageable arrays and with page-locked ones (pinned_empty / ofarn_host_alloc).  Never the headline `value`
(bench.py keeps inputs resident in HBM); reported in DESIGN.md 7.  (Measured and not kept: sending wave i's results to the host on a
transfer stream while wave i+1 is computed -- 36.3 ms for 64 pairs in waves of 16 against 34.8 ms as one wave: on this platform the
device-to-host copy is a blit kernel that fills the chip, so it does not run beside the next wave's kernels.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pairs
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    W, H = 1920, 1080
    fr, _ = translated_pairs(4, H, W, 3000)
    frames = np.concatenate([fr] * (n // 4), 0)
    with ofa.FarnebackEngine(W, H, n, 0, levels=5) as eng:
        eng.calc_batch(frames[:8])
        for want_flow in (True, False):
            t0 = time.perf_counter()
            eng.calc_batch(frames, want_flow=want_flow)
            dt = time.perf_counter() - t0
            print(f"host-pointer batch of {n} pairs, flow copied back: {want_flow}: {dt * 1e3:.1f} ms -> {n / dt:.0f} pairs/s "
                  f"(device part {eng.last_device_ms:.1f} ms)")
        # the same with page-locked buffers on both sides (ofarn_host_alloc / pinned_empty)
        pin = ofa.pinned_empty(frames.shape, np.uint8)
        pin[...] = frames
        out = ofa.pinned_empty((n, H, W, 2), np.float32)
        eng.calc_batch(pin[:8], out_flow=out[:4])
        for rep in range(2):
            t0 = time.perf_counter()
            eng.calc_batch(pin, out_flow=out)
            dt = time.perf_counter() - t0
            print(f"host-pointer batch of {n} pairs, page-locked frames and flow (buffers allocated once): {dt * 1e3:.1f} ms -> {n / dt:.0f} pairs/s "
                  f"(device part {eng.last_device_ms:.1f} ms)")


if __name__ == "__main__":
    main()
