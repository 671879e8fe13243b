"""The viewers' per-frame device path with the composited frame coming back: a decoded BGR frame in, the frame with one of the flow layers
added onto it out (DenseOF.py:574-582, pathfinder_viewer.py:297-300), at 1920x1080 with the function's default levels=3.

    python3 tools/viewloop.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hackathonopticalflow_amd as H
from examples.headless_viewer import synthetic_video
w, h = 1920, 1080
video = synthetic_video(12, h, w)
def loop(fn, n=120):
    with H.FlowStream(levels=3) as st:
        for i in range(20):
            fn(st, video[i % 12])
        t0 = time.perf_counter()
        for i in range(n):
            fn(st, video[i % 12])
        return (time.perf_counter() - t0) / n * 1e3
def arrows(st, img):
    if st.next_view(img, danger=False, arrows=None) is not None:
        st.view_arrows(14, over_frame=True)
def rainbow(st, img):
    if st.next_view(img, danger=False, arrows=None) is not None:
        st.view_rainbow(over_frame=True)
def lamps(st, img):
    if st.next_view(img, danger=True, arrows=None) is not None:
        st.view_lamps(over_frame=True)
def allthree(st, img):
    if st.next_view(img, danger=True, arrows=14) is not None:
        st.view_lamps(over_frame=True)
print(f"BGR frame in -> frame + draw_flow arrows out: {loop(arrows):.3f} ms per 1080p frame")
print(f"BGR frame in -> frame + draw_hsv rainbow out: {loop(rainbow):.3f} ms")
print(f"BGR frame in -> frame + obstacle lamps out:   {loop(lamps):.3f} ms")
print(f"BGR frame in -> danger map + arrow lines + frame with lamps out: {loop(allthree):.3f} ms")
