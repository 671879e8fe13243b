#!/usr/bin/env python3
"""One wave of N pairs against two waves of N/2 on the library's two internal streams (per-kernel timing off), same box, alternating;
the caller's stream is torch's legacy null stream or a non-blocking stream of its own.

    python tools/wave_ab.py [--batch 512] [--reps 5] [--rounds 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import torch
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pairs
    W, H, B = 1920, 1080, a.batch
    dev = torch.device("cuda", 0)
    fr, _ = translated_pairs(4, H, W, 3000)
    fr = torch.from_numpy(fr).to(dev)
    frames = torch.empty((2 * B, H, W), dtype=torch.uint8, device=dev)
    for i in range(B):
        frames[2 * i:2 * i + 2] = fr[2 * (i % 4):2 * (i % 4) + 2]
    flow = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
    P = len(ofa.grid_points(W, H, 30))
    mask = torch.zeros((B, P), dtype=torch.uint8, device=dev)
    v = torch.zeros_like(mask)
    own = torch.cuda.Stream(device=dev)
    engs = {"one wave": ofa.FarnebackEngine(W, H, B, 0, levels=5), "two waves, two internal streams": ofa.FarnebackEngine(W, H, B // 2, 0, levels=5)}
    single = ofa.FarnebackEngine(W, H, B // 2, 0, levels=5)
    single.set_option("single_stream", 1)
    engs["two waves, one stream"] = single
    torch.cuda.synchronize()
    for rnd in range(a.rounds):
        for sname, st in (("null stream", torch.cuda.default_stream(dev)), ("own non-blocking stream", own)):
            for name, eng in engs.items():
                run = lambda: eng.calc_batch_device(frames, 2 * B, W, H, ofa.PAIRS_INDEPENDENT, flow, mask, v, stream=st.cuda_stream)
                run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    run()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.reps
                print(f"round {rnd + 1}: caller on {sname:24s} {name:34s} {dt * 1e3:8.3f} ms per {B} pairs -> {B / dt:8.1f} pairs/s", flush=True)


if __name__ == "__main__":
    main()
