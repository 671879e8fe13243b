import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hackathonopticalflow_amd as H
rng = np.random.default_rng(3)
base = rng.integers(0, 256, (1080 + 64, 1920 + 64), dtype=np.uint8)
frames = [np.ascontiguousarray(base[i % 32:i % 32 + 1080, (3 * i) % 32:(3 * i) % 32 + 1920]) for i in range(16)]
for levels in (5, 3):
    with H.FarnebackEngine(1920, 1080, 1, levels=levels) as eng:
        out = H.pinned_empty((1080, 1920, 2))
        for name, fn in (("calc_reuse, unrelated pairs (pair turn)", lambda i: eng.calc_reuse(frames[(5 * i) & 15], frames[(5 * i + 3) & 15], out)),
                         ("calc (ofarn_calc), same pairs", lambda i: eng.calc(frames[(5 * i) & 15], frames[(5 * i + 3) & 15], out)),
                         ("calc_reuse, loop pattern (hits)", lambda i: eng.calc_reuse(frames[i & 15], frames[(i + 1) & 15], out))):
            for i in range(10):
                fn(i)
            ts, dev = [], []
            for i in range(60):
                t0 = time.perf_counter(); fn(i); ts.append((time.perf_counter() - t0) * 1e3); dev.append(eng.last_device_ms)
            print(f"levels={levels} {name:42s} wall median {np.median(ts):.3f} ms (min {min(ts):.3f}), device {np.median(dev):.3f}  reuse {eng.reuse_info()}", flush=True)
