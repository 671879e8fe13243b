#!/usr/bin/env python3
"""The frame loop of the reference's viewers without the GUI, on the MI355X path.

What DenseOF.py:491-525 and pathfinder_viewer.py do per frame (decode, cvtColor, calcOpticalFlowFarneback or
calcOpticalFlowPyrLK, vector filter, draw_hsv / draw_flow / draw_sparse_lamps) is run here on a stack of BGR frames
that is already on the GPU; what comes back per frame pair is small: the danger mask and V of the grid points, the
int32 arrow end points, optionally the HSV rainbow.  Drawing the overlays (cv2.polylines / cv2.circle) stays with
the caller.

    python examples/headless_viewer.py [--frames 9] [--width 640 --height 480]

Synthetic input (a textured scene panning by an integer shift per frame); real video decoding is out of scope.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_video(n, h, w, seed=1):
    """BGR frames uint8[n,h,w,3] of one smooth random scene panning by (2, 1) px per frame."""
    from hackathonopticalflow_amd.synth import translated_pair
    pad = 4 * n + 16
    base, _, _ = translated_pair(h + 2 * pad, w + 2 * pad, seed, max_shift=0)
    frames = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        g = base[pad - i:pad - i + h, pad - 2 * i:pad - 2 * i + w]
        frames[i] = np.stack([g, np.roll(g, 3, axis=1), 255 - g], -1)
    return frames


def run(frames_bgr, device=0, step=30, want_hsv=True):
    """Returns a dict of per-pair results for the consecutive pairs of the stack (all NumPy, small)."""
    import torch
    import hackathonopticalflow_amd as ofa
    n, h, w, _ = frames_bgr.shape
    n_pairs = n - 1
    dev = torch.device("cuda", device)
    d_bgr = torch.from_numpy(frames_bgr).to(dev)
    pts = ofa.grid_points(w, h, step)
    P = len(pts)
    K = ofa.load_library().ofarn_flow_arrow_count(w, h, 14, None, None)
    d_flow = torch.empty((n_pairs, h, w, 2), dtype=torch.float32, device=dev)
    d_mask = torch.zeros((n_pairs, P), dtype=torch.uint8, device=dev)
    d_v = torch.zeros_like(d_mask)
    d_lines = torch.zeros((n_pairs, K, 2, 2), dtype=torch.int32, device=dev)
    d_rainbow = torch.empty((n_pairs, h, w, 3), dtype=torch.uint8, device=dev) if want_hsv else None
    d_gray = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    d_pts = torch.from_numpy(pts).to(dev)
    d_next = torch.zeros((n_pairs, P, 2), dtype=torch.float32, device=dev)
    d_st = torch.zeros((n_pairs, P), dtype=torch.uint8, device=dev)
    d_err = torch.zeros((n_pairs, P), dtype=torch.float32, device=dev)
    d_lk_mask = torch.zeros_like(d_mask)
    d_lk_v = torch.zeros_like(d_mask)
    d_lk_flow = torch.zeros((n_pairs, P, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    with ofa.FarnebackEngine(w, h, max(1, min(n_pairs, 64)), device, grid_step=step) as eng:      # DenseOF.py defaults
        # dense path: DenseOF.py:510 (cvtColor) + :520 (Farneback) + the grid filter, video order (prev_gray = gray)
        eng.calc_batch_device(d_bgr, n, w, h, ofa.PAIRS_CONSECUTIVE, d_flow, d_mask, d_v, stream=st, bgr=True)
        eng.flow_arrows_device(d_flow, n_pairs, w, h, 14, d_lines, stream=st)                      # draw_flow's lines
        if want_hsv:
            eng.flow_hsv_device(d_flow, n_pairs, w, h, None, d_rainbow, stream=st)                 # draw_hsv
        # sparse path of pathfinder_viewer.py: LK from the later frame back to the earlier one, then the same filter
        eng.bgr2gray_device(d_bgr, n, w, h, d_gray, stream=st)
        eng.lk_batch_device(d_gray, n, w, h, ofa.PAIRS_CONSECUTIVE, d_pts, P, d_next, d_st, d_err, reverse=True, stream=st,
                            winSize=(45, 45), maxLevel=2, criteria=(3, 10, 0.03))
        eng.vector_filter_device(d_next - d_pts, n_pairs, w, h, d_lk_mask, d_lk_v, d_lk_flow, stream=st)
        torch.cuda.synchronize()
    return dict(points=pts, dense_mask=d_mask.cpu().numpy(), dense_v=d_v.cpu().numpy(), lines=d_lines.cpu().numpy(),
                rainbow=None if d_rainbow is None else d_rainbow.cpu().numpy(), lk_mask=d_lk_mask.cpu().numpy(),
                lk_v=d_lk_v.cpu().numpy(), lk_flow=d_lk_flow.cpu().numpy(), lk_status=d_st.cpu().numpy(),
                mean_flow=d_flow.mean(dim=(1, 2)).cpu().numpy())


def run_loop(frames_bgr, device=0, step=30):
    """The same dense results produced the way the reference's loop produces them -- one decoded BGR frame per turn
    (DenseOF.py:491-525), host arrays in and out -- through FlowStream: the previous frame stays on the device, each turn
    uploads one frame.  Returns (flows float32[n-1,H,W,2], masks uint8[n-1,P], vs uint8[n-1,P])."""
    import hackathonopticalflow_amd as ofa
    flows, masks, vs = [], [], []
    with ofa.FlowStream(device=device, copy=True) as stream:                  # DenseOF.py defaults
        for img in frames_bgr:
            flow = stream.next(img)                                             # cvtColor + Farneback; None for the first frame
            if flow is None:
                continue
            mask, v = ofa.danger_map(flow, step, device=device)                 # pathfinder_viewer.py:159-176, 204-217
            flows.append(flow); masks.append(mask); vs.append(v)
    return np.stack(flows), np.stack(masks), np.stack(vs)


def run_loop_view(frames_bgr, device=0, step=30):
    """The loop again, taking per frame only what the viewers DRAW (danger map, draw_flow's arrow lines, the obstacle layer added
    onto the frame as pathfinder_viewer.py:299-300 does): FlowStream.next_view keeps the 16.6 MB flow field on the device.
    Returns (masks, vs, lines, composited frames)."""
    import hackathonopticalflow_amd as ofa
    masks, vs, lines, shown = [], [], [], []
    with ofa.FlowStream(device=device, grid_step=step) as stream:
        for img in frames_bgr:
            view = stream.next_view(img, danger=True, arrows=14)
            if view is None:
                continue
            masks.append(view["mask"].copy()); vs.append(view["v"].copy()); lines.append(view["lines"].copy())
            shown.append(stream.view_lamps(over_frame=True).copy())       # output_bgr = cv2.add(img, draw_sparse_lamps(...))
    return np.stack(masks), np.stack(vs), np.stack(lines), np.stack(shown)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=9)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    a = ap.parse_args()
    video = synthetic_video(a.frames, a.height, a.width)
    out = run(video)
    _, loop_masks, _ = run_loop(video)
    print("frame loop (FlowStream) and batch danger masks identical:", bool((loop_masks == out["dense_mask"]).all()))
    view_masks, _, view_lines, shown = run_loop_view(video)
    print("view loop (flow stays on the GPU): masks identical:", bool((view_masks == out["dense_mask"]).all()),
          " arrow lines identical:", bool((view_lines == out["lines"]).all()),
          " pixels the obstacle layer changed per frame:", [int((shown[i] != video[i + 1]).any(-1).sum()) for i in range(len(shown))])
    for i in range(a.frames - 1):
        print(f"pair {i}: mean dense flow {out['mean_flow'][i].round(3)}  danger points dense {int(out['dense_mask'][i].sum())}"
              f" / LK {int(out['lk_mask'][i].sum())} of {len(out['points'])}  LK tracked {int(out['lk_status'][i].sum())}")


if __name__ == "__main__":
    main()
