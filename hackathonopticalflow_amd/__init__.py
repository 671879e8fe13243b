"""hackathonopticalflow_amd -- MI355X-native dense Farneback optical flow + danger-point filter.

Drop-in for the one hot path of spirinis/HackathonOpticalFlow (DenseOF.py:127-157 ->
cv2.calcOpticalFlowFarneback; pathfinder_viewer.py:159-176 vector filter), plus the frame front end
(BGR -> gray) and the dense visualisers either side of it.  See DESIGN.md.
"""
from .ofarn import (FILTER_DENSEOF, FILTER_VIEWER, FlowStream, MultiGpuEngine, shard_pairs_c, pinned_empty, OPTFLOW_FARNEBACK_GAUSSIAN, OPTFLOW_USE_INITIAL_FLOW,  # noqa: F401
                    FarnebackEngine, OfarnLkParams, OfarnParams, PAIRS_CONSECUTIVE, PAIRS_INDEPENDENT,
                    calcOpticalFlowFarneback, calcOpticalFlowPyrLK, get_flow_lk, make_lk_params, calculate_optical_flow, close_cached_engines, cvtColor_bgr2gray,
                    danger_map, draw_flow, draw_hsv, draw_sparse_lamps, flow_lines, grid_points, level_plan, load_library, make_params)

__all__ = ["FILTER_DENSEOF", "FILTER_VIEWER", "FlowStream", "MultiGpuEngine", "shard_pairs_c", "pinned_empty", "OPTFLOW_FARNEBACK_GAUSSIAN", "OPTFLOW_USE_INITIAL_FLOW",
           "FarnebackEngine", "OfarnLkParams", "OfarnParams", "PAIRS_CONSECUTIVE", "PAIRS_INDEPENDENT",
           "calcOpticalFlowFarneback", "calcOpticalFlowPyrLK", "get_flow_lk", "make_lk_params", "calculate_optical_flow", "close_cached_engines", "cvtColor_bgr2gray",
           "danger_map", "draw_flow", "draw_hsv", "draw_sparse_lamps", "flow_lines", "grid_points", "level_plan", "load_library", "make_params"]
