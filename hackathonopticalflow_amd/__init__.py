"""hackathonopticalflow_amd -- MI355X-native dense Farneback optical flow + danger-point filter.

Drop-in for the one hot path of spirinis/HackathonOpticalFlow (DenseOF.py:127-157 ->
cv2.calcOpticalFlowFarneback; pathfinder_viewer.py:159-176 vector filter).  See DESIGN.md.
"""
from .ofarn import (FILTER_DENSEOF, FILTER_VIEWER, FarnebackEngine, OfarnParams, PAIRS_CONSECUTIVE,  # noqa: F401
                    PAIRS_INDEPENDENT,
                    calcOpticalFlowFarneback, calculate_optical_flow, close_cached_engines, danger_map,
                    grid_points, level_plan, load_library, make_params)

__all__ = ["FILTER_DENSEOF", "FILTER_VIEWER", "FarnebackEngine", "OfarnParams", "PAIRS_CONSECUTIVE", "PAIRS_INDEPENDENT",
           "calcOpticalFlowFarneback", "calculate_optical_flow", "close_cached_engines", "danger_map",
           "grid_points", "level_plan", "load_library", "make_params"]
