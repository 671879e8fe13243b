#!/usr/bin/env python3
"""Small driver for profiling: runs the hot path on one wave of synthetic 1080p pairs.

    python tools/kbench.py [--levels 5] [--batch 64] [--reps 3] [--w 1920 --h 1080]

Prints the per-(stage, level) hipEvent table.  Meant to sit behind `rocprofv3 ... -- python3 tools/kbench.py`.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--no-danger", action="store_true")
    ap.add_argument("--winsize", type=int, default=15)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--poly-n", type=int, default=5)
    ap.add_argument("--box-order", type=int, default=0, help="1: the box window summed in OpenCV's literal order (verification mode)")
    a = ap.parse_args()
    import torch
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pairs
    dev = torch.device("cuda", 0)
    uniq = min(4, a.batch)
    fr, _ = translated_pairs(uniq, a.h, a.w, 3000)
    fr = torch.from_numpy(fr).to(dev)
    frames = torch.empty((2 * a.batch, a.h, a.w), dtype=torch.uint8, device=dev)
    for i in range(a.batch):
        frames[2 * i:2 * i + 2] = fr[2 * (i % uniq):2 * (i % uniq) + 2]
    flow = torch.empty((a.batch, a.h, a.w, 2), dtype=torch.float32, device=dev)
    P = len(ofa.grid_points(a.w, a.h, 30))
    mask = torch.zeros((a.batch, P), dtype=torch.uint8, device=dev)
    v = torch.zeros_like(mask)
    eng = ofa.FarnebackEngine(a.w, a.h, a.batch, 0, levels=a.levels, iterations=a.iterations, winsize=a.winsize, flags=a.flags,
                              poly_n=a.poly_n, poly_sigma=1.2 if a.poly_n == 5 else 1.5)
    if a.box_order:
        eng.set_option("box_order", 1)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: eng.calc_batch_device(frames, 2 * a.batch, a.w, a.h, ofa.PAIRS_INDEPENDENT, flow,
                                        None if a.no_danger else mask, None if a.no_danger else v, stream=st)
    run()
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    rows = eng.profile_read()
    tot = sum(r["ms"] for r in rows)
    for r in sorted(rows, key=lambda r: -r["ms"]):
        print(f"{r['stage']:16s} L{r['level']} n={r['launches']:3d} avg={r['ms'] / r['launches']:8.4f} ms  "
              f"{r['units'] / r['launches'] / (r['ms'] / r['launches']) / 1e6:8.2f} Gunit/s")
    print(f"total {tot / a.reps:.3f} ms per pass of {a.batch} pairs -> {a.batch / (tot / a.reps) * 1e3:.1f} pairs/s")
    # wall clock of the same passes with per-kernel timing off (kernels on internal streams overlap: the sum above then overstates)
    eng.profile_enable(False)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(f"wall {dt * 1e3:.3f} ms per pass of {a.batch} pairs -> {a.batch / dt:.1f} pairs/s")


if __name__ == "__main__":
    main()
