"""A bare streaming loop for a kernel trace: N device-resident turns at 1920x1080 (levels 5), nothing else in the process.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/turn_trace -- python3 tools/turn_loop.py 40
    python3 tools/turn_trace.py gpurun_out/turn_trace
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hackathonopticalflow_amd as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
view = len(sys.argv) > 2 and sys.argv[2] == "view"
w, h = 1920, 1080
rng = np.random.default_rng(5)
base = rng.integers(0, 256, (h + 64, w + 64), dtype=np.uint8)
frames = torch.from_numpy(np.stack([base[i:i + h, 2 * i:2 * i + w] for i in range(8)])).cuda()
P = len(H.grid_points(w, h, 30))
flow = torch.empty((h, w, 2), dtype=torch.float32, device="cuda")
mask = torch.empty(P, dtype=torch.uint8, device="cuda")
v = torch.empty(P, dtype=torch.uint8, device="cuda")
with H.FarnebackEngine(w, h, 1, levels=5) as eng:
    for i in range(n):
        eng.stream_next_device(frames[i & 7], w, h, flow, mask, v)
        torch.cuda.synchronize()
print("done", float(flow.abs().mean()))
