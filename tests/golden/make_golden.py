"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/farneback_oracle.c + the NumPy filter).

    python tests/golden/make_golden.py

The reference ships no fixtures for its Farneback path and cv2 cannot be imported here (SURVEY.md
8c), so these vectors are outputs of the repo's own oracle, PARITY UNPINNED against real OpenCV.
They pin (a) the oracle against accidental change and (b) the HIP path on the GPU box, where they
are compared bit for bit (flow_direct: the device summation order) and within tolerance (flow_running = OpenCV's literal order).
flow_direct_row_ltr: flow_direct with the oracle's OFO_ROW_SMALL_SYMM switch off (the order of rounds 1-2, still selectable on both sides).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hackathonopticalflow_amd.synth import translated_pair, warped_pair  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    # name: (height, width, seed, farneback kwargs)
    "g160x120_L2": (120, 160, 501, dict(levels=2)),
    "g97x83_odd": (83, 97, 502, dict(levels=1, winsize=9, poly_n=7, poly_sigma=1.5)),
    "g192x144_scale08": (144, 192, 503, dict(levels=3, pyr_scale=0.8, iterations=2, winsize=8)),
    # FPV-like family: zoom + rotation + sub-pixel shift + occluding patch (synth.warped_pair)
    "g200x150_warp": (150, 200, 504, dict(levels=2)),
}


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    for name, (h, w, seed, kw) in CASES.items():
        if name.endswith("_warp"):
            prev, nxt, _, _ = warped_pair(h, w, seed)
            shift = (0, 0)
        else:
            prev, nxt, shift = translated_pair(h, w, seed, max_shift=4)
        fd = O.farneback(prev, nxt, box_mode=O.BOX_BLOCKED, **kw)
        fr = O.farneback(prev, nxt, box_mode=O.BOX_RUNNING, **kw)
        mask, v = O.danger_map_numpy(fd, w, h, 30)
        # the same with the row pass of 3- and 5-tap Gaussian kernels left to right (OFO_ROW_SMALL_SYMM = 0, the rounds 1-2 order)
        O.set_row_small_symm(False)
        try:
            fl = O.farneback(prev, nxt, box_mode=O.BOX_BLOCKED, **kw)
        finally:
            O.set_row_small_symm(True)
        np.savez_compressed(os.path.join(out, name + ".npz"), prev=prev, next=nxt, shift=np.int64(shift),
                            flow_direct=fd, flow_running=fr, flow_direct_row_ltr=fl, mask=mask, v=v,
                            params=np.array(repr(sorted(kw.items()))))
        print(name, fd.shape, "max |direct-running| =", float(np.abs(fd - fr).max()), "kept", int(mask.sum()))


if __name__ == "__main__":
    main()
