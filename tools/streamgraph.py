#!/usr/bin/env python3
"""Experiment: device time of a streaming turn (one new 1080p frame, device-resident) launched eagerly vs replayed from
HIP graphs (two graphs: the R slots alternate).  Decides whether the library should replay graphs in its host entry points.

    python tools/streamgraph.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pair
    w, h = 1920, 1080
    f0, f1, _ = translated_pair(h, w, 2001)
    d = [torch.from_numpy(f).cuda() for f in (f0, f1)]
    flow = torch.empty((h, w, 2), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    st = side.cuda_stream
    for overlap in (1, 2, 0, 2, 1):
        eng = ofa.FarnebackEngine(w, h, 1, 0, levels=5)
        eng.set_option("stream_overlap", overlap)
        with torch.cuda.stream(side):
            eng.stream_next_device(d[0], w, h, flow, stream=st)
            for i in range(6):
                eng.stream_next_device(d[(i + 1) % 2], w, h, flow, stream=st)
            side.synchronize()
            # eager timing
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for i in range(20):
                e0.record(side)
                eng.stream_next_device(d[i % 2], w, h, flow, stream=st)
                e1.record(side)
                side.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"overlap={overlap} eager : device {np.median(ts) * 1e3:.1f} us (min {min(ts) * 1e3:.1f})")
            graphs = []
            for i in range(2):     # 20 turns done: the next turn reads frame d[0] into the slot parity of an even turn
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    eng.stream_next_device(d[i % 2], w, h, flow, stream=st)
                graphs.append(g)
            ts = []
            for i in range(20):
                e0.record(side)
                graphs[i % 2].replay()
                e1.record(side)
                side.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"overlap={overlap} graphs: device {np.median(ts) * 1e3:.1f} us (min {min(ts) * 1e3:.1f})")
            ref = ofa.FarnebackEngine(w, h, 1, 0, levels=5)
            want = ref.calc(f0, f1)
            ok = np.array_equal(flow.cpu().numpy(), want)
            print("   last replay (pair f0 -> f1) bit-exact vs ofarn_calc:", ok)
            ref.close()
            del graphs
        eng.close()


if __name__ == "__main__":
    main()
