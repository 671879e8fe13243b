"""CPU ORACLE bindings -- test infrastructure, NOT product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  ``hackathonopticalflow_amd`` never does.

Two things live here:

* ctypes bindings to ``oracle/_build/libofarn_oracle*.so`` (farneback_oracle.c / filter_oracle.c),
  the plain-C restatement of OpenCV 4.10's CPU Farneback that the reference reaches through
  ``cv2.calcOpticalFlowFarneback`` (DenseOF.py:147-156).  PARITY UNPINNED -- see the C header.
* the reference's own NumPy lines for the measurement grid, the vector filter and the danger
  brightness, re-typed from pathfinder_viewer.py:159-176, 204-217, 252-267 and executed by the
  real NumPy, so their float32/float64 promotion semantics are the reference's.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

BOX_RUNNING = 0  # OpenCV's literal running-sum order
BOX_DIRECT = 1   # direct fixed-order window sums
BOX_BLOCKED = 2  # block-restarted running column sums in double (the order the HIP kernels use)


class OfoParams(C.Structure):
    _fields_ = [("pyr_scale", C.c_double), ("levels", C.c_int), ("winsize", C.c_int),
                ("iterations", C.c_int), ("poly_n", C.c_int), ("poly_sigma", C.c_double),
                ("flags", C.c_int)]


class OfoCapture(C.Structure):
    _fields_ = [(n, C.POINTER(C.POINTER(C.c_float)))
                for n in ("I0", "I1", "R0", "R1", "M_first", "flow_init", "flow_out")]


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (oracle/Makefile)."""
    so = os.path.join(_BUILD, "libofarn_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("farneback_oracle.c", "filter_oracle.c", "frontend_oracle.c", "lk_oracle.c", "Makefile")]
    if (not force and os.path.exists(so)
            and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs)
            and os.path.exists(os.path.join(_BUILD, "libofarn_oracle_omp.so"))):
        return
    subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)


_libs: dict[bool, C.CDLL] = {}


def lib(omp: bool = False) -> C.CDLL:
    if omp not in _libs:
        build()
        l = C.CDLL(os.path.join(_BUILD, "libofarn_oracle_omp.so" if omp else "libofarn_oracle.so"))
        fp, u8p, ip, dp = (C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int),
                           C.POINTER(C.c_double))
        l.ofo_crop_levels.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        l.ofo_crop_levels.restype = C.c_int
        l.ofo_level_geom.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, ip, ip, dp, ip]
        l.ofo_level_geom.restype = None
        l.ofo_gaussian_kernel.argtypes = [C.c_int, C.c_double, fp]
        l.ofo_gaussian_blur.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_double, fp]
        l.ofo_gaussian_blur.restype = None
        l.ofo_resize_linear.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        l.ofo_resize_linear.restype = None
        l.ofo_level_image.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                      C.c_int, C.c_int, fp]
        l.ofo_level_image.restype = None
        l.ofo_poly_prepare.argtypes = [C.c_int, C.c_double, fp, fp, fp, dp]
        l.ofo_poly_prepare.restype = None
        l.ofo_polyexp.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_double, fp]
        l.ofo_polyexp.restype = None
        l.ofo_update_matrices.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        l.ofo_update_matrices.restype = None
        l.ofo_update_flow_blur.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        l.ofo_update_flow_blur.restype = None
        l.ofo_update_flow_gaussian.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        l.ofo_update_flow_gaussian.restype = None
        l.ofo_farneback_ex.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.POINTER(OfoParams),
                                       C.c_int, fp, C.POINTER(OfoCapture)]
        l.ofo_farneback_ex.restype = C.c_int
        l.ofo_farneback_batch.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(OfoParams), C.c_int, fp, C.c_int]
        l.ofo_farneback_batch.restype = C.c_int
        l.ofo_max_threads.restype = C.c_int
        l.ofo_set_row_small_symm.argtypes = [C.c_int]
        l.ofo_set_row_small_symm.restype = None
        l.ofo_get_row_small_symm.restype = C.c_int
        l.ofo_grid_points.argtypes = [C.c_int, C.c_int, C.c_int, fp]
        l.ofo_grid_points.restype = C.c_int
        l.ofo_vector_filter.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, u8p, fp,
                                        C.POINTER(C.c_int32), u8p, dp]
        l.ofo_vector_filter.restype = C.c_int
        l.ofo_vector_filter2.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, u8p, fp,
                                         C.POINTER(C.c_int32), u8p, dp]
        l.ofo_vector_filter2.restype = C.c_int
        l.ofo_gray_coeffs.argtypes = [C.c_int, ip, ip, ip, ip]
        l.ofo_gray_coeffs.restype = None
        l.ofo_bgr2gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        l.ofo_bgr2gray.restype = None
        l.ofo_area_tab.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, fp]
        l.ofo_area_tab.restype = C.c_int
        l.ofo_resize_area.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        l.ofo_resize_area.restype = C.c_int
        l.ofo_hsv2bgr_u8.argtypes = [u8p, C.c_size_t, u8p]
        l.ofo_hsv2bgr_u8.restype = None
        l.ofo_pyrdown_u8.argtypes = [u8p, C.c_int, C.c_int, u8p]
        l.ofo_pyrdown_u8.restype = None
        l.ofo_scharr_deriv.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(C.c_int16)]
        l.ofo_scharr_deriv.restype = None
        l.ofo_lk_levels.argtypes = [C.c_int] * 5
        l.ofo_lk_levels.restype = C.c_int
        l.ofo_pyr_lk.argtypes = [u8p, u8p, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                 C.c_int, C.c_double, C.c_int, fp, u8p, fp]
        l.ofo_pyr_lk.restype = C.c_int
        _libs[omp] = l
    return _libs[omp]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _params(pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2, flags=0):
    return OfoParams(float(pyr_scale), int(levels), int(winsize), int(iterations), int(poly_n),
                     float(poly_sigma), int(flags))


def set_row_small_symm(on: bool) -> bool:
    """OFO_ROW_SMALL_SYMM switch of farneback_oracle.c (both builds of the library): True (default) = the GaussianBlur row
    pass of a ksize <= 5 kernel in SymmRowSmallFilter's order, False = plain left to right (rounds 1-2).  Returns the
    previous setting."""
    prev = bool(lib().ofo_get_row_small_symm())
    for omp in (False, True):
        lib(omp).ofo_set_row_small_symm(int(bool(on)))
    return prev


# --------------------------------------------------------------------------- geometry
def crop_levels(W, H, pyr_scale, levels) -> int:
    return lib().ofo_crop_levels(W, H, pyr_scale, levels)


def level_geom(W, H, pyr_scale, k):
    w, h, ks, sg = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    lib().ofo_level_geom(W, H, pyr_scale, k, C.byref(w), C.byref(h), C.byref(sg), C.byref(ks))
    return w.value, h.value, sg.value, ks.value


# --------------------------------------------------------------------------- stages
def gaussian_kernel(n, sigma):
    out = np.empty(n, np.float32)
    lib().ofo_gaussian_kernel(n, sigma, _fp(out))
    return out


def gaussian_blur(img_f32, ksize, sigma):
    src = np.ascontiguousarray(img_f32, np.float32)
    dst = np.empty_like(src)
    lib().ofo_gaussian_blur(_fp(src), src.shape[1], src.shape[0], ksize, sigma, _fp(dst))
    return dst


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.float32)
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.float32)
    lib().ofo_resize_linear(_fp(src), src.shape[1], src.shape[0], cn, _fp(dst), dw, dh)
    return dst


def level_image(img_u8, ksize, sigma, w, h):
    img = np.ascontiguousarray(img_u8, np.uint8)
    out = np.empty((h, w), np.float32)
    lib().ofo_level_image(_u8p(img), img.shape[1], img.shape[0], img.shape[1], ksize, sigma, w, h, _fp(out))
    return out


def poly_prepare(n, sigma):
    g = np.zeros(2 * n + 1, np.float32)
    xg = np.zeros_like(g)
    xxg = np.zeros_like(g)
    ig = np.zeros(4, np.float64)
    off = n * 4  # bytes to the centre element
    cast = lambda a: C.cast(a.ctypes.data + off, C.POINTER(C.c_float))
    lib().ofo_poly_prepare(n, sigma, cast(g), cast(xg), cast(xxg), ig.ctypes.data_as(C.POINTER(C.c_double)))
    return g, xg, xxg, ig


def polyexp(I, n=5, sigma=1.2):
    I = np.ascontiguousarray(I, np.float32)
    R = np.empty(I.shape + (5,), np.float32)
    lib().ofo_polyexp(_fp(I), I.shape[1], I.shape[0], n, sigma, _fp(R))
    return R


def update_matrices(R0, R1, flow):
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    M = np.empty((h, w, 5), np.float32)
    lib().ofo_update_matrices(_fp(R0), _fp(R1), _fp(flow), _fp(M), w, h, 0, h)
    return M


def update_flow_blur(R0, R1, flow, M, winsize, update_matrices_flag, box_mode=BOX_RUNNING):
    """Returns (new_flow, new_M); inputs are not modified."""
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.array(flow, np.float32, order="C", copy=True)
    M = np.array(M, np.float32, order="C", copy=True)
    h, w = flow.shape[:2]
    lib().ofo_update_flow_blur(_fp(R0), _fp(R1), _fp(flow), _fp(M), w, h, winsize,
                               int(bool(update_matrices_flag)), box_mode)
    return flow, M


def update_flow_gaussian(R0, R1, flow, M, winsize, update_matrices_flag):
    """FarnebackUpdateFlow_GaussianBlur (OPTFLOW_FARNEBACK_GAUSSIAN).  Returns (new_flow, new_M)."""
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.array(flow, np.float32, order="C", copy=True)
    M = np.array(M, np.float32, order="C", copy=True)
    h, w = flow.shape[:2]
    lib().ofo_update_flow_gaussian(_fp(R0), _fp(R1), _fp(flow), _fp(M), w, h, winsize, int(bool(update_matrices_flag)))
    return flow, M


@dataclass
class Capture:
    I0: list
    I1: list
    R0: list
    R1: list
    M_first: list
    flow_init: list
    flow_out: list


def farneback(prev, next, pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5,
              poly_sigma=1.2, flags=0, box_mode=BOX_RUNNING, capture=False, init_flow=None):
    """CPU oracle for cv2.calcOpticalFlowFarneback(prev, next, flow, ...) -> float32[H,W,2].

    flags & 4 (OPTFLOW_USE_INITIAL_FLOW): init_flow float32[H,W,2] is cv2's in/out `flow` on entry.
    With capture=True also returns a Capture of per-level intermediates (index = level k)."""
    prev = np.ascontiguousarray(prev, np.uint8)
    next = np.ascontiguousarray(next, np.uint8)
    assert prev.ndim == 2 and prev.shape == next.shape
    H, W = prev.shape
    p = _params(pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)
    flow = np.empty((H, W, 2), np.float32)
    if flags & 4:
        if init_flow is None or np.shape(init_flow) != (H, W, 2):
            raise ValueError("OPTFLOW_USE_INITIAL_FLOW needs init_flow float32[H,W,2]")
        flow = np.array(init_flow, np.float32, order="C")
    cap_struct = None
    cap = None
    if capture:
        nl = crop_levels(W, H, pyr_scale, levels) + 1
        geo = [level_geom(W, H, pyr_scale, k) for k in range(nl)]
        mk = lambda ch: [np.zeros((g[1], g[0]) + ((ch,) if ch > 1 else ()), np.float32) for g in geo]
        cap = Capture(mk(1), mk(1), mk(5), mk(5), mk(5), mk(2), mk(2))
        cap_struct = OfoCapture()
        keep = []
        for name in ("I0", "I1", "R0", "R1", "M_first", "flow_init", "flow_out"):
            arr = (C.POINTER(C.c_float) * nl)(*[_fp(a) for a in getattr(cap, name)])
            keep.append(arr)
            setattr(cap_struct, name, C.cast(arr, C.POINTER(C.POINTER(C.c_float))))
    rc = lib().ofo_farneback_ex(_u8p(prev), _u8p(next), W, H, W, C.byref(p), box_mode, _fp(flow),
                                C.byref(cap_struct) if capture else None)
    if rc != 0:
        raise ValueError(f"oracle farneback rejected arguments (rc={rc})")
    return (flow, cap) if capture else flow


def farneback_batch(frames, pairs_mode=0, nthreads=1, box_mode=BOX_RUNNING, **kw):
    """frames u8[n_frames,H,W] -> flow float32[n_pairs,H,W,2]; OpenMP across pairs if nthreads != 1."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape
    n_pairs = n - 1 if pairs_mode else n // 2
    p = _params(**kw)
    flow = np.empty((n_pairs, H, W, 2), np.float32)
    rc = lib(omp=(nthreads != 1)).ofo_farneback_batch(_u8p(frames), n_pairs, pairs_mode, W, H,
                                                       C.byref(p), box_mode, _fp(flow), nthreads)
    if rc != 0:
        raise ValueError(f"oracle farneback_batch failed (rc={rc})")
    return flow


def omp_max_threads() -> int:
    return lib(omp=True).ofo_max_threads()


# --------------------------------------------------------------------------- filter: C twin
def grid_points_c(width, height, step=30):
    P = lib().ofo_grid_points(width, height, step, None)
    pts = np.empty((P, 2), np.float32)
    lib().ofo_grid_points(width, height, step, _fp(pts))
    return pts


def vector_filter_c(vec, pts, width, height, variant=0):
    vec = np.ascontiguousarray(vec, np.float32)
    pts = np.ascontiguousarray(pts, np.float32)
    P = len(pts)
    mask = np.empty(P, np.uint8)
    mod = np.empty(P, np.float32)
    iflow = np.empty((P, 2), np.int32)
    v = np.empty(P, np.uint8)
    thr = np.empty(2, np.float64)
    lib().ofo_vector_filter2(_fp(vec), _fp(pts), P, width, height, variant, _u8p(mask), _fp(mod),
                            iflow.ctypes.data_as(C.POINTER(C.c_int32)), _u8p(v),
                            thr.ctypes.data_as(C.POINTER(C.c_double)))
    return mask.astype(bool), mod, iflow, v, thr


# --------------------------------------------------------------------------- filter: the reference's NumPy
def grid_points_numpy(width, height, step=30):
    """pathfinder_viewer.py:255-267, re-typed."""
    if width // step % 2 == 1:
        indent_w = width % step / 2
    else:
        indent_w = (width % step + step) / 2
    if height // step % 2 == 1:
        indent_h = height % step / 2
    else:
        indent_h = (height % step + step) / 2
    points_grid = np.mgrid[indent_w:width:step, indent_h:height:step].astype(int)
    points = []
    for x, y in zip(points_grid[0].flatten(), points_grid[1].flatten()):
        points.append([x, y])
    return np.array(points).astype(np.float32).reshape(-1, 2)


def _cr(fn):
    """Correctly rounded float32 version of a NumPy transcendental: evaluated in float64, rounded once.  NumPy's own
    float32 arctan2 / cos / sin are SIMD approximations (up to 3.2 ulp for arctan2 on this container's build) that
    differ between CPUs; the HIP kernels and the C oracle are held to the correctly rounded value instead."""
    def f(*a):
        return fn(*[np.asarray(x, np.float64) for x in a]).astype(np.float32)
    return f


def vector_filter_numpy(flow_, points_, width, height, variant=0, cr=True):
    """pathfinder_viewer.py:159-176, re-typed (flow_ = next_pts - points_ is the input here).
    variant=1 uses the older gate of DenseOF.py:228 instead of pathfinder_viewer.py:173.
    cr=True evaluates arctan2 / cos / sin correctly rounded (see _cr); cr=False runs the lines literally,
    i.e. with whatever this machine's NumPy float32 loops return.

    Returns mask bool[P], modulus float32[P], int flow int32[P,2] for ALL points
    (reference keeps [mask]), int points int32[P,2]."""
    half_width = int(width / 2)
    half_height = int(height / 2)
    fx, fy = flow_[:, 0], flow_[:, 1]
    x, y = points_[:, 0], points_[:, 1]
    arctan2, cos, sin = (_cr(np.arctan2), _cr(np.cos), _cr(np.sin)) if cr else (np.arctan2, np.cos, np.sin)
    ang = arctan2(fy, fx)
    modulus = np.sqrt(fx * fx + fy * fy)
    modulus_middle = np.sqrt((half_width - x) ** 2 + (half_height - y) ** 2)
    modulus = modulus / (5 + np.sqrt(modulus_middle)) * 30
    fx = modulus * cos(ang)
    fy = modulus * sin(ang)
    next_pts = np.vstack([x + fx, y + fy]).T
    next_pts = np.int32(next_pts + 0.5)
    ipoints = np.int32(points_ + 0.5)
    if variant == 1:
        mask = np.greater(modulus, np.median(modulus) * 1.2)                                   # DenseOF.py:228
    else:
        mask = (np.median(modulus) * 1.0 < modulus) & (modulus < np.percentile(modulus, 99))   # pathfinder_viewer.py:173
    return mask, modulus, next_pts - ipoints, ipoints


def lamp_values_numpy(iflow_kept):
    """pathfinder_viewer.py:204-217: V channel written per kept point (uint8 store truncates)."""
    fx, fy = iflow_kept[:, 0], iflow_kept[:, 1]
    modulus = np.sqrt(fx * fx + fy * fy)
    out = np.zeros(len(iflow_kept), np.uint8)
    for i, m in enumerate(modulus):
        out[i] = np.minimum(50 + m * 2, 255)
    return out


def danger_map_numpy(flow_hw2, width, height, step=30, variant=0, return_flow=False, cr=True):
    """Dense adaptation: sample flow[y,x] at the grid (DenseOF.py:44-45), then filter.

    Returns (mask u8[P], v u8[P]) with v = 0 at rejected points."""
    pts = grid_points_numpy(width, height, step)
    xi = pts[:, 0].astype(np.int64)
    yi = pts[:, 1].astype(np.int64)
    vec = np.ascontiguousarray(flow_hw2[yi, xi, :], np.float32)
    mask, _, iflow, _ = vector_filter_numpy(vec, pts, width, height, variant, cr=cr)
    v = np.zeros(len(pts), np.uint8)
    if mask.any():
        v[mask] = lamp_values_numpy(iflow[mask])
    if return_flow:
        return mask.astype(np.uint8), v, iflow
    return mask.astype(np.uint8), v


# --------------------------------------------------------------------------- front end and visualisers
GRAY_15BIT = 0  # OpenCV >= 3.4.2 / 4.x
GRAY_14BIT = 1  # older releases


def bgr2gray(img_bgr, variant=GRAY_15BIT):
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) on uint8[H,W,3] (DenseOF.py:481,510); C restatement."""
    a = np.ascontiguousarray(img_bgr, np.uint8)
    h, w, _ = a.shape
    out = np.empty((h, w), np.uint8)
    lib().ofo_bgr2gray(_u8p(a), w, h, 3 * w, variant, _u8p(out))
    return out


def area_tab(ssize, dsize):
    cap = 2 * ssize + 2
    si, di = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    al = np.zeros(cap, np.float32)
    ip = C.POINTER(C.c_int)
    n = lib().ofo_area_tab(ssize, dsize, cap, si.ctypes.data_as(ip), di.ctypes.data_as(ip), _fp(al))
    return si[:n], di[:n], al[:n]


def resize_area(src, dw, dh):
    """cv::resize(src, (dw, dh), INTER_AREA), float32, shrinking only."""
    src = np.ascontiguousarray(src, np.float32)
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.float32)
    rc = lib().ofo_resize_area(_fp(src), src.shape[1], src.shape[0], cn, _fp(dst), dw, dh)
    if rc:
        raise ValueError("resize_area: only shrinking is restated")
    return dst


def hsv2bgr_u8(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) on uint8[...,3]; C restatement of HSV2RGB_b."""
    a = np.ascontiguousarray(hsv, np.uint8)
    out = np.empty_like(a)
    lib().ofo_hsv2bgr_u8(_u8p(a), a.size // 3, _u8p(out))
    return out


def draw_hsv_planes_numpy(flow_, cr=True):
    """DenseOF.py:109-120, re-typed: the uint8 HSV image draw_hsv hands to cv2.cvtColor.  cr: see vector_filter_numpy."""
    h, w = flow_.shape[:2]
    fx, fy = flow_[:, :, 0], flow_[:, :, 1]

    ang = (_cr(np.arctan2) if cr else np.arctan2)(fy, fx) + np.pi
    v = np.sqrt(fx * fx + fy * fy)

    hsv = np.zeros((h, w, 3), np.uint8)
    hsv[..., 0] = ang * (180 / np.pi / 2)
    hsv[..., 1] = 255
    hsv[..., 2] = np.minimum(v * 4, 255)
    return hsv


def draw_hsv_numpy(flow_, cr=True):
    """draw_hsv (DenseOF.py:109-124): NumPy lines by the real NumPy, HSV2BGR by the C restatement."""
    return hsv2bgr_u8(draw_hsv_planes_numpy(flow_, cr=cr))


def cv_circle_filled(img, center, radius, color, fill=True):
    """cv2.circle(img, center, radius, color, thickness=-1), lineType LINE_8, shift 0: drawing.cpp Circle() -- a midpoint loop that
    fills the spans [cx-dx, cx+dx] on rows cy+-dy and [cx-dy, cx+dy] on rows cy+-dx, clipped to the image.  In place.
    fill=False: thickness=1, the same loop putting only the span end points (the 8 symmetric points of each step)."""
    h, w = img.shape[:2]
    cx, cy = int(center[0]), int(center[1])

    def hline(y, x1, x2):
        if 0 <= y < h:
            if not fill:
                for x in (x1, x2):
                    if 0 <= x < w:
                        img[y, x] = color
                return
            x1, x2 = max(x1, 0), min(x2, w - 1)
            if x1 <= x2:
                img[y, x1:x2 + 1] = color

    err, dx, dy, plus, minus = 0, int(radius), 0, 1, (int(radius) << 1) - 1
    while dx >= dy:
        hline(cy - dy, cx - dx, cx + dx)
        hline(cy + dy, cx - dx, cx + dx)
        hline(cy - dx, cx - dy, cx + dy)
        hline(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        mask = (1 if err <= 0 else 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2


def cv_clip_line(width, height, pt1, pt2):
    """cv::clipLine(Size2l, Point2l&, Point2l&) (drawing.cpp): -> (visible, pt1, pt2).  64-bit integers, the intersection through a
    double division truncated toward zero."""
    x1, y1, x2, y2 = int(pt1[0]), int(pt1[1]), int(pt2[0]), int(pt2[1])
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, (x1, y1), (x2, y2)
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * float(x2 - x1) / float(y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * float(x2 - x1) / float(y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * float(y2 - y1) / float(x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * float(y2 - y1) / float(x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def cv_line8(img, pt1, pt2, color):
    """cv2.line / one segment of cv2.polylines with thickness 1, LINE_8, shift 0: drawing.cpp Line() = clipLine when an end point lies
    outside the image, then LineIterator(pt1, pt2, connectivity 8, leftToRight=true) -- Bresenham whose major axis advances every
    step and whose minor axis advances when the error term is negative.  In place."""
    h, w = img.shape[:2]
    x1, y1, x2, y2 = int(pt1[0]), int(pt1[1]), int(pt2[0]), int(pt2[1])
    if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
        ok, (x1, y1), (x2, y2) = cv_clip_line(w, h, (x1, y1), (x2, y2))
        if not ok:
            return
    dx, dy = x2 - x1, y2 - y1
    delta_x = delta_y = 1
    px, py = x1, y1
    if dx < 0:                       # leftToRight
        dx, dy = -dx, -dy
        px, py = x2, y2
    if dy < 0:
        dy, delta_y = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    major = (0, delta_y) if vert else (delta_x, 0)
    minor = (delta_x, 0) if vert else (0, delta_y)
    for _ in range(dx + 1):
        img[py, px] = color
        neg = err < 0
        err += minus_delta + (plus_delta if neg else 0)
        px += major[0] + (minor[0] if neg else 0)
        py += major[1] + (minor[1] if neg else 0)


def cv_polylines(img, pts, is_closed, color):
    """cv2.polylines(img, pts, isClosed, color) with the default thickness 1 / LINE_8 (drawing.cpp PolyLine)."""
    for poly in pts:
        poly = np.asarray(poly).reshape(-1, 2)
        if len(poly) == 0:
            continue
        p0 = poly[-1] if is_closed else poly[0]
        for p in poly[(0 if is_closed else 1):]:
            cv_line8(img, p0, p, color)
            p0 = p


def draw_flow_numpy(img_shape, flow, step=14):
    """draw_flow (DenseOF.py:40-59 = pathfinder_viewer.py:51-73), re-typed: NumPy lines by the real NumPy, cv2.polylines and
    cv2.circle by the restatements above."""
    h, w = img_shape
    img_bgr = np.zeros((h, w, 3), np.uint8)
    y, x = np.mgrid[step / 2:h:step, step / 2:w:step].reshape(2, -1).astype(int)
    fx, fy = flow[y, x].T

    lines = np.vstack([x, y, x - fx, y - fy]).T.reshape(-1, 2, 2)
    lines = np.int32(lines + 0.5)

    cv_polylines(img_bgr, lines, False, (0, 255, 0))

    for (x1, y1), (_x2, _y2) in lines:
        cv_circle_filled(img_bgr, (x1, y1), 1, (0, 255, 0))

    return img_bgr


def draw_sparse_lamps_numpy(flow_, points_, width, height, radius=6):
    """draw_sparse_lamps (pathfinder_viewer.py:196-222), re-typed: NumPy lines by the real NumPy, HSV2BGR and cv2.circle by the
    restatements above.  flow_, points_: the kept int32 vectors and points get_flow_lk returns (:175-176)."""
    fx, fy = flow_[:, 0], flow_[:, 1]
    ang = np.arctan2(fy, fx) + np.pi
    modulus = np.sqrt(fx * fx + fy * fy)

    hsv = np.zeros((height, width, 3), np.uint8)
    for (x, y), a, m in zip(points_, ang, modulus):
        hsv[y, x, 0] = 0
        hsv[y, x, 1] = 255
        hsv[y, x, 2] = np.minimum(50 + m * 2, 255)

    bgr = hsv2bgr_u8(hsv)
    for x, y, in points_:
        cv_circle_filled(bgr, (x, y), radius, (int(bgr[y, x, 0]), int(bgr[y, x, 1]), int(bgr[y, x, 2])))
    return bgr


def cv_add_u8(a, b):
    """cv2.add on uint8 arrays: saturating (pathfinder_viewer.py:299-300)."""
    return np.minimum(a.astype(np.uint16) + b.astype(np.uint16), 255).astype(np.uint8)


def draw_flow_lines_numpy(img_shape, flow, step=14):
    """DenseOF.py:40-49, re-typed: the int32 `lines` array handed to cv2.polylines."""
    h, w = img_shape
    y, x = np.mgrid[step / 2:h:step, step / 2:w:step].reshape(2, -1).astype(int)
    fx, fy = flow[y, x].T
    lines = np.vstack([x, y, x - fx, y - fy]).T.reshape(-1, 2, 2)
    lines = np.int32(lines + 0.5)
    return lines


# --------------------------------------------------------------------------- sparse pyramidal Lucas-Kanade
LK_SUM_SCALAR = 0    # row-major float accumulation: lkpyramid.cpp's scalar loop
LK_SUM_COLUMNS = 1   # per window column, then the columns left to right: the HIP kernel's order
LK_GET_MIN_EIGENVALS = 8
LK_USE_INITIAL_FLOW = 4


def pyrdown_u8(img):
    """cv2.pyrDown on uint8[H,W] (BORDER_REFLECT_101, as buildOpticalFlowPyramid calls it)."""
    a = np.ascontiguousarray(img, np.uint8)
    h, w = a.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().ofo_pyrdown_u8(_u8p(a), w, h, _u8p(out))
    return out


def scharr_deriv(img):
    """calcScharrDeriv: int16[H,W,2] = (dI/dx, dI/dy), unscaled 3-10-3 Scharr."""
    a = np.ascontiguousarray(img, np.uint8)
    h, w = a.shape
    out = np.empty((h, w, 2), np.int16)
    lib().ofo_scharr_deriv(_u8p(a), w, h, out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def lk_levels(w, h, win, max_level):
    return lib().ofo_lk_levels(w, h, win[0], win[1], max_level)


def calc_optical_flow_pyr_lk(prev, next, pts, next_pts=None, winSize=(21, 21), maxLevel=3, criteria=(30, 0.01), flags=0,
                             minEigThreshold=1e-4, sum_mode=LK_SUM_SCALAR):
    """cv2.calcOpticalFlowPyrLK(prev, next, pts, next_pts, winSize=, maxLevel=, criteria=(COUNT|EPS, count, eps), flags=,
    minEigThreshold=) -> (next_pts float32[n,2], status uint8[n], err float32[n]); criteria here = (count, eps)."""
    prev = np.ascontiguousarray(prev, np.uint8)
    next = np.ascontiguousarray(next, np.uint8)
    pts = np.ascontiguousarray(np.asarray(pts, np.float32).reshape(-1, 2))
    n = len(pts)
    out = np.zeros((n, 2), np.float32) if next_pts is None else np.array(np.asarray(next_pts, np.float32).reshape(-1, 2), order="C")
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    h, w = prev.shape
    rc = lib().ofo_pyr_lk(_u8p(prev), _u8p(next), w, h, _fp(pts), n, winSize[0], winSize[1], maxLevel, criteria[0],
                          criteria[1], flags, minEigThreshold, sum_mode, _fp(out), _u8p(status), _fp(err))
    if rc:
        raise ValueError(f"oracle pyr_lk rejected arguments (rc={rc})")
    return out, status, err


def get_flow_lk_numpy(img1, img2, points_, width, height, sum_mode=LK_SUM_SCALAR, next_pts=None):
    """pathfinder_viewer.py:144-178 (get_flow_lk) without the drawing: LK from img2 to img1 at the grid points, then
    the vector filter.  Returns (mask bool[P], flow int32[P,2] for all points, ipoints int32[P,2], next_pts float32[P,2]).
    next_pts: use these tracked points instead of running the oracle LK (to compare filters on identical vectors)."""
    if next_pts is None:
        next_pts, _status, _err = calc_optical_flow_pyr_lk(img2, img1, points_, None, winSize=(45, 45), maxLevel=2,
                                                            criteria=(10, 0.03), sum_mode=sum_mode)
    flow_ = next_pts - points_
    mask, _mod, iflow, ipts = vector_filter_numpy(flow_, points_, width, height, 0)
    return mask, iflow, ipts, next_pts


def get_flow_lk_layer_numpy(mask, iflow, ipts, width, height, draw_bad_flow=False):
    """The drawing part of get_flow_lk (pathfinder_viewer.py:147, 177-192), re-typed, on the filter's results for ALL points (mask
    bool[P], iflow = next_pts - points_ int32[P,2], ipts int32[P,2]): cv2.polylines / cv2.circle by the restatements above."""
    frame_layer = np.zeros((height, width, 3), np.uint8)
    mask = np.asarray(mask, bool)
    next_all = ipts + iflow
    points_, points_bad = ipts[mask], ipts[~mask]
    next_pts, nextPts_bad = next_all[mask], next_all[~mask]
    lines = np.concatenate((points_, next_pts), axis=1)
    rlines = lines.reshape(-1, 2, 2)
    cv_polylines(frame_layer, rlines, False, (0, 0, 255))
    for x1, y1, _x2, _y2 in lines:
        cv_circle_filled(frame_layer, (x1, y1), 1, (255, 0, 255), fill=False)
    if draw_bad_flow:
        lines_bad = np.concatenate((points_bad, nextPts_bad), axis=1)
        rlines_bad = lines_bad.reshape(-1, 2, 2)
        cv_polylines(frame_layer, rlines_bad, False, (255, 255, 0))
        for x1, y1, _x2, _y2 in lines_bad:
            cv_circle_filled(frame_layer, (x1, y1), 1, (255, 255, 0), fill=False)
    return frame_layer
