"""Pins for the CPU oracle (oracle/farneback_oracle.c) against independent closed forms.

The reference has no tests or golden vectors for its Farneback call, and cv2 is
not importable here, so the oracle is PARITY UNPINNED against real OpenCV.  These checks pin each
stage of the restatement against scipy / numpy float64 closed forms instead.
"""
import numpy as np
import pytest
from scipy import ndimage

from hackathonopticalflow_amd.synth import translated_pair


def test_level_geometry_table(oracle):
    # half-to-even rounding, sigma, ksize
    exp1080 = [(1920, 1080, 0.0, 3), (960, 540, 0.5, 3), (480, 270, 1.5, 9), (240, 135, 3.5, 19),
               (120, 68, 7.5, 39), (60, 34, 15.5, 79)]
    for k, e in enumerate(exp1080):
        assert oracle.level_geom(1920, 1080, 0.5, k) == e
    assert oracle.level_geom(3840, 2160, 0.5, 6) == (60, 34, 31.5, 159)
    assert oracle.crop_levels(1920, 1080, 0.5, 5) == 5
    assert oracle.crop_levels(3840, 2160, 0.5, 6) == 6
    assert oracle.crop_levels(640, 480, 0.5, 3) == 3
    # min_size = 32 crops: 640x480 at scale 1/16 is 40x30 -> 30 < 32 stops at k=3
    assert oracle.crop_levels(640, 480, 0.5, 10) == 3
    assert oracle.crop_levels(64, 64, 0.5, 5) == 1


@pytest.mark.parametrize("n,sigma", [(3, 0.5), (9, 1.5), (19, 3.5), (39, 7.5), (79, 15.5), (159, 31.5)])
def test_gaussian_kernel_closed_form(oracle, n, sigma):
    x = np.arange(n) - (n - 1) / 2
    w = np.exp(-0.5 * x * x / sigma ** 2)
    w /= w.sum()
    got = oracle.gaussian_kernel(n, sigma)
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, w, rtol=2e-7, atol=0)
    assert abs(float(got.astype(np.float64).sum()) - 1) < 1e-6


def test_gaussian_kernel_fixed_table(oracle):
    np.testing.assert_array_equal(oracle.gaussian_kernel(3, 0.0), np.float32([0.25, 0.5, 0.25]))


@pytest.mark.parametrize("ksize,sigma", [(3, 0.0), (3, 0.5), (9, 1.5), (39, 7.5)])
def test_gaussian_blur_vs_scipy_mirror(oracle, ksize, sigma):
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 255, (70, 93)).astype(np.float32)
    k = oracle.gaussian_kernel(ksize, sigma).astype(np.float64)
    ref = ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror")
    ref = ndimage.correlate1d(ref, k, axis=0, mode="mirror")
    got = oracle.gaussian_blur(img, ksize, sigma)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4)


def test_resize_exact_2x_is_block_mean(oracle):
    rng = np.random.default_rng(6)
    img = rng.uniform(0, 255, (40, 64)).astype(np.float32)
    got = oracle.resize_linear(img, 32, 20)
    blk = ((img[0::2, 0::2] + img[0::2, 1::2]) + (img[1::2, 0::2] + img[1::2, 1::2])) * np.float32(0.25)
    np.testing.assert_array_equal(got, blk)  # bit exact: all weights are 0.5


def test_resize_exact_4x_is_centre_mean(oracle):
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 255, (64, 128)).astype(np.float32)
    got = oracle.resize_linear(img, 32, 16)
    c = img.reshape(16, 4, 32, 4)[:, 1:3, :, 1:3].astype(np.float64).mean(axis=(1, 3))
    np.testing.assert_allclose(got, c, rtol=1e-6)


def _resize_ref64(src, dw, dh):
    sh, sw = src.shape[:2]

    def coords(d, s):
        f = ((np.arange(d) + 0.5) * (1.0 / (d / s)) - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i).astype(np.float32)
        lo = i < 0
        i[lo], f[lo] = 0, 0
        hi = i >= s - 1
        i[hi], f[hi] = s - 1, 0
        return i, np.minimum(i + 1, s - 1), f.astype(np.float64)

    x0, x1, fx = coords(dw, sw)
    y0, y1, fy = coords(dh, sh)
    s = src.astype(np.float64)
    if s.ndim == 2:
        s = s[..., None]
    fx = fx[None, :, None]
    fy = fy[:, None, None]
    top = s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx
    out = top * (1 - fy) + bot * fy
    return out if src.ndim == 3 else out[..., 0]


@pytest.mark.parametrize("sw,sh,dw,dh,cn", [(1920 // 8, 1080 // 8, 120, 68, 1), (120, 68, 240, 135, 2),
                                            (60, 34, 120, 68, 2), (97, 53, 33, 41, 1)])
def test_resize_general_vs_float64(oracle, sw, sh, dw, dh, cn):
    rng = np.random.default_rng(8)
    src = rng.uniform(-20, 255, (sh, sw) if cn == 1 else (sh, sw, cn)).astype(np.float32)
    got = oracle.resize_linear(src, dw, dh)
    np.testing.assert_allclose(got, _resize_ref64(src, dw, dh), rtol=0, atol=1e-4)


def _gram(n, sigma):
    x = np.arange(-n, n + 1)
    g = np.exp(-x * x / (2 * sigma * sigma)).astype(np.float32).astype(np.float64)
    g = (g / g.sum()).astype(np.float32).astype(np.float64)
    X, Y = np.meshgrid(x, x)
    Wt = np.outer(g, g)
    basis = [np.ones_like(X), X, Y, X * X, Y * Y, X * Y]
    G = np.array([[np.sum(Wt * a * b) for b in basis] for a in basis], np.float64)
    return g, G


@pytest.mark.parametrize("n,sigma", [(5, 1.2), (7, 1.5), (5, 1.1), (3, 0.9)])
def test_inverse_gram_constants(oracle, n, sigma):
    g, G = _gram(n, sigma)
    inv = np.linalg.inv(G)
    og, oxg, oxxg, ig = oracle.poly_prepare(n, sigma)
    np.testing.assert_allclose(og, g, rtol=1e-7)
    x = np.arange(-n, n + 1)
    np.testing.assert_allclose(oxg, x * g, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(oxxg, x * x * g, rtol=1e-6, atol=1e-12)
    # optflowgf.cpp forms g[y]*g[x]*x*x in float before accumulating in double: ~1e-7 relative
    np.testing.assert_allclose(ig, [inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]], rtol=1e-6)


def test_polyexp_reproduces_quadratic(oracle):
    # I = a + b x + c y + d x^2 + e y^2 + f xy  ->  interior R = [c', b', e, d, f] where the linear
    # terms are those of the local expansion about each pixel.
    h, w = 40, 48
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    a, b, c, d, e, f = 3.0, 0.5, -0.25, 0.03125, -0.015625, 0.0078125
    I = a + b * x + c * y + d * x * x + e * y * y + f * x * y
    R = oracle.polyexp(I.astype(np.float32), 5, 1.2)
    s = np.s_[6:-6, 6:-6]
    np.testing.assert_allclose(R[s][..., 0], (c + 2 * e * y + f * x)[s], atol=2e-4)
    np.testing.assert_allclose(R[s][..., 1], (b + 2 * d * x + f * y)[s], atol=2e-4)
    np.testing.assert_allclose(R[s][..., 2], e, atol=2e-5)
    np.testing.assert_allclose(R[s][..., 3], d, atol=2e-5)
    np.testing.assert_allclose(R[s][..., 4], f, atol=2e-5)


def test_polyexp_vs_weighted_lsq_float64(oracle):
    """Independent statement: R = (G^-1 B^T W I)[1..5] per pixel, replicate borders."""
    n, sigma = 5, 1.2
    rng = np.random.default_rng(11)
    I = ndimage.gaussian_filter(rng.standard_normal((37, 45)), 1.5).astype(np.float32) * 50
    g, G = _gram(n, sigma)
    inv = np.linalg.inv(G)
    x = np.arange(-n, n + 1)
    k0, k1, k2 = g, x * g, x * x * g
    P = np.pad(I.astype(np.float64), n, mode="edge")

    def corr(ky, kx):
        t = ndimage.correlate1d(P, ky, axis=0, mode="constant")
        t = ndimage.correlate1d(t, kx, axis=1, mode="constant")
        return t[n:-n, n:-n]

    m = {"1": corr(k0, k0), "x": corr(k0, k1), "y": corr(k1, k0), "xx": corr(k0, k2),
         "yy": corr(k2, k0), "xy": corr(k1, k1)}
    mom = np.stack([m["1"], m["x"], m["y"], m["xx"], m["yy"], m["xy"]], -1)
    coef = mom @ inv.T   # basis order 1, x, y, xx, yy, xy
    ref = np.stack([coef[..., 2], coef[..., 1], coef[..., 4], coef[..., 3], coef[..., 5]], -1)
    got = oracle.polyexp(I, n, sigma)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4)


def _update_matrices_ref64(R0, R1, flow):
    h, w = flow.shape[:2]
    R0 = R0.astype(np.float64)
    R1 = R1.astype(np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    dx = flow[..., 0].astype(np.float64)
    dy = flow[..., 1].astype(np.float64)
    fx = (xx.astype(np.float32) + flow[..., 0]).astype(np.float64)
    fy = (yy.astype(np.float32) + flow[..., 1]).astype(np.float64)
    x1 = np.floor(fx).astype(np.int64)
    y1 = np.floor(fy).astype(np.int64)
    ax = fx - x1
    ay = fy - y1
    inb = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc = np.clip(x1, 0, w - 2)
    yc = np.clip(y1, 0, h - 2)
    samp = ((1 - ax) * (1 - ay))[..., None] * R1[yc, xc] + (ax * (1 - ay))[..., None] * R1[yc, xc + 1] \
        + ((1 - ax) * ay)[..., None] * R1[yc + 1, xc] + (ax * ay)[..., None] * R1[yc + 1, xc + 1]
    r2 = np.where(inb, samp[..., 0], 0)
    r3 = np.where(inb, samp[..., 1], 0)
    r4 = np.where(inb, (R0[..., 2] + samp[..., 2]) * 0.5, R0[..., 2])
    r5 = np.where(inb, (R0[..., 3] + samp[..., 3]) * 0.5, R0[..., 3])
    r6 = np.where(inb, (R0[..., 4] + samp[..., 4]) * 0.25, R0[..., 4] * 0.5)
    r2 = (R0[..., 0] - r2) * 0.5 + r4 * dy + r6 * dx
    r3 = (R0[..., 1] - r3) * 0.5 + r6 * dy + r5 * dx
    tab = np.float32([0.14, 0.14, 0.4472, 0.4472, 0.4472]).astype(np.float64)
    sx = np.ones(w)
    sx[:5] *= tab
    sx[-5:] *= tab[::-1]
    sy = np.ones(h)
    sy[:5] *= tab
    sy[-5:] *= tab[::-1]
    s = sy[:, None] * sx[None, :]
    r2, r3, r4, r5, r6 = (v * s for v in (r2, r3, r4, r5, r6))
    return np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3,
                     r6 * r2 + r5 * r3], -1)


def test_update_matrices_vs_float64(oracle):
    rng = np.random.default_rng(12)
    h, w = 41, 57
    R0 = rng.standard_normal((h, w, 5)).astype(np.float32)
    R1 = rng.standard_normal((h, w, 5)).astype(np.float32)
    flow = (rng.standard_normal((h, w, 2)) * 3).astype(np.float32)
    flow[0, 0] = (-5, -5)          # out of bounds branch
    flow[h - 1, w - 1] = (4.5, 0.25)
    got = oracle.update_matrices(R0, R1, flow)
    ref = _update_matrices_ref64(R0, R1, flow)
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5)


@pytest.mark.parametrize("winsize", [15, 7, 8, 3])
@pytest.mark.parametrize("mode", [0, 1])
def test_box_blur_and_solve_vs_scipy(oracle, winsize, mode):
    rng = np.random.default_rng(13)
    h, w = 45, 52
    A = rng.standard_normal((h, w, 2, 2))
    r4, r5, r6 = A[..., 0, 0], A[..., 1, 1], A[..., 0, 1]
    r2, r3 = rng.standard_normal((2, h, w))
    M = np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3,
                  r6 * r2 + r5 * r3], -1).astype(np.float32)
    m = winsize // 2
    taps = 2 * m + 1
    blur = np.stack([ndimage.uniform_filter(M[..., c].astype(np.float64), taps, mode="nearest")
                     for c in range(5)], -1) * (taps * taps) / float(winsize * winsize)
    Gm = np.empty((h, w, 2, 2))
    Gm[..., 0, 0], Gm[..., 0, 1], Gm[..., 1, 0], Gm[..., 1, 1] = blur[..., 0], blur[..., 1], blur[..., 1], blur[..., 2]
    det = blur[..., 0] * blur[..., 2] - blur[..., 1] ** 2 + 1e-3
    # flow = [dx, dy]:  dx = (g11*h2 - g12*h1)/det,  dy = (g22*h1 - g12*h2)/det
    ref = np.stack([(blur[..., 0] * blur[..., 4] - blur[..., 1] * blur[..., 3]) / det,
                    (blur[..., 2] * blur[..., 3] - blur[..., 1] * blur[..., 4]) / det], -1)
    zeros = np.zeros((h, w, 5), np.float32)
    got, _ = oracle.update_flow_blur(zeros, zeros, np.zeros((h, w, 2), np.float32), M, winsize, False, mode)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_solve_is_regularised_normal_equations(oracle):
    # on a constant M the window average is M itself: the step solves G [dy,dx]^T = h up to the
    # +1e-3 on the determinant
    g11, g12, g22, h1, h2 = 2.0, 0.5, 1.5, 0.7, -0.3
    M = np.tile(np.float32([g11, g12, g22, h1, h2]), (33, 33, 1))
    z5 = np.zeros((33, 33, 5), np.float32)
    got, _ = oracle.update_flow_blur(z5, z5, np.zeros((33, 33, 2), np.float32), M, 3, False, 0)
    G = np.array([[g11, g12], [g12, g22]], np.float64)
    sol = np.linalg.solve(G, [h1, h2]) * (np.linalg.det(G) / (np.linalg.det(G) + 1e-3))
    np.testing.assert_allclose(got[16, 16], [sol[1], sol[0]], rtol=1e-6)


def test_gaussian_window_vs_scipy(oracle):
    """OPTFLOW_FARNEBACK_GAUSSIAN: separable Gaussian (sigma = 0.3*(winsize/2)), replicate borders."""
    rng = np.random.default_rng(2)
    h, w, ws = 45, 52, 15
    A = rng.standard_normal((h, w, 2, 2))
    r4, r5, r6 = A[..., 0, 0], A[..., 1, 1], A[..., 0, 1]
    r2, r3 = rng.standard_normal((2, h, w))
    M = np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3], -1).astype(np.float32)
    m = ws // 2
    x = np.arange(-m, m + 1)
    k = np.exp(-x * x / (2 * (m * 0.3) ** 2))
    k /= k.sum()
    bl = np.stack([ndimage.correlate1d(ndimage.correlate1d(M[..., c].astype(np.float64), k, axis=0, mode="nearest"),
                                       k, axis=1, mode="nearest") for c in range(5)], -1)
    det = bl[..., 0] * bl[..., 2] - bl[..., 1] ** 2 + 1e-3
    ref = np.stack([(bl[..., 0] * bl[..., 4] - bl[..., 1] * bl[..., 3]) / det,
                    (bl[..., 2] * bl[..., 3] - bl[..., 1] * bl[..., 4]) / det], -1)
    z5 = np.zeros_like(M)
    got, _ = oracle.update_flow_gaussian(z5, z5, np.zeros((h, w, 2), np.float32), M, ws, False)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)
    p, n, (tx, ty) = translated_pair(240, 320, 9)
    f = oracle.farneback(p, n, levels=2, flags=256)
    assert np.linalg.norm(f[40:-40, 40:-40] - np.float32([tx, ty]), axis=-1).mean() < 0.15


def test_constant_image_gives_zero_flow(oracle):
    img = np.full((64, 80), 77, np.uint8)
    flow = oracle.farneback(img, img, levels=1)
    assert flow.shape == (64, 80, 2) and flow.dtype == np.float32
    assert np.all(flow == 0)


def test_identical_frames_give_zero_flow(oracle):
    p, _, _ = translated_pair(192, 256, 3)
    flow = oracle.farneback(p, p, levels=2)
    # the last row/column take the out-of-bounds branch of UpdateMatrices (x1 == w-1), so the
    # window-averaged solve is only exactly zero away from the bottom/right edge
    # ... and that edge effect spreads winsize/2 px per iteration and doubles per level
    assert np.abs(flow[:40, :60]).max() < 1e-4


@pytest.mark.parametrize("seed", [1001, 1002])
def test_translation_ground_truth_640x480(oracle, seed):
    # BASELINE config 1 shape with DenseOF.py:127-128 defaults (levels=3)
    p, n, (tx, ty) = translated_pair(480, 640, seed)
    flow = oracle.farneback(p, n)
    epe = np.linalg.norm(flow[32:-32, 32:-32] - np.float32([tx, ty]), axis=-1)
    assert epe.mean() < 0.1, (tx, ty, epe.mean())


def test_box_summation_orders_agree(oracle):
    """OpenCV's literal running sums, plain direct sums and the block-restarted running sums the
    HIP kernels use are the same window in a different order of double additions."""
    p, n, _ = translated_pair(200, 260, 21)
    a = oracle.farneback(p, n, levels=2, box_mode=oracle.BOX_RUNNING)
    b = oracle.farneback(p, n, levels=2, box_mode=oracle.BOX_DIRECT)
    c = oracle.farneback(p, n, levels=2, box_mode=oracle.BOX_BLOCKED)
    for other in (b, c):
        d = np.linalg.norm(a - other, axis=-1)
        assert d.mean() < 1e-5 and d.max() < 1e-3
    d = np.linalg.norm(b - c, axis=-1)
    assert d.max() < 1e-6


@pytest.mark.parametrize("winsize", [2, 3, 8, 15, 41])
def test_blocked_column_sums_equal_direct_window(oracle, winsize):
    # values with a short mantissa: every double sum is exact, so the orders must agree bit for bit
    rng = np.random.default_rng(15)
    h, w = 67, 45
    M = (rng.integers(-512, 512, (h, w, 5)) / 8.0).astype(np.float32)
    z5, z2 = np.zeros_like(M), np.zeros((h, w, 2), np.float32)
    a, _ = oracle.update_flow_blur(z5, z5, z2, M, winsize, False, oracle.BOX_DIRECT)
    b, _ = oracle.update_flow_blur(z5, z5, z2, M, winsize, False, oracle.BOX_BLOCKED)
    np.testing.assert_array_equal(a, b)


def test_capture_matches_stage_functions(oracle):
    p, n, _ = translated_pair(100, 132, 22)
    flow, cap = oracle.farneback(p, n, levels=1, capture=True)
    for k in (0, 1):
        w, h, sigma, ks = oracle.level_geom(132, 100, 0.5, k)
        np.testing.assert_array_equal(cap.I0[k], oracle.level_image(p, ks, sigma, w, h))
        np.testing.assert_array_equal(cap.R1[k], oracle.polyexp(cap.I1[k]))
        np.testing.assert_array_equal(cap.M_first[k], oracle.update_matrices(cap.R0[k], cap.R1[k], cap.flow_init[k]))
    np.testing.assert_array_equal(cap.flow_out[0], flow)
    assert np.all(cap.flow_init[1] == 0)
    up = oracle.resize_linear(cap.flow_out[1], 132, 100) * np.float32(2)
    np.testing.assert_array_equal(cap.flow_init[0], up)


def test_golden_fixtures_pin_the_oracle(oracle):
    """tests/golden/*.npz were written by tests/golden/make_golden.py from this oracle; they keep it
    from drifting and are what the GPU box compares the HIP path against."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert len(files) >= 3
    for path in files:
        g = np.load(path, allow_pickle=False)
        kw = dict(eval(str(g["params"])))
        np.testing.assert_array_equal(oracle.farneback(g["prev"], g["next"], box_mode=oracle.BOX_BLOCKED, **kw),
                                      g["flow_direct"])
        np.testing.assert_array_equal(oracle.farneback(g["prev"], g["next"], box_mode=oracle.BOX_RUNNING, **kw),
                                      g["flow_running"])
        h, w = g["prev"].shape
        mask, v = oracle.danger_map_numpy(g["flow_direct"], w, h, 30)
        np.testing.assert_array_equal(mask, g["mask"])
        np.testing.assert_array_equal(v, g["v"])


def test_row_small_symm_switch(oracle):
    """OFO_ROW_SMALL_SYMM (OpenCV's SymmRowSmallFilter order for 3- and 5-tap Gaussian rows, the default) against the plain
    left-to-right order of rounds 1-2: identical where the arithmetic is exact (level 0: [1/4, 1/2, 1/4] on byte values, and any
    kernel wider than 5 taps, which never takes that filter), different in the last bits at a sigma = 0.5 level (ksize 3) and at
    a ksize-5 level; both match a direct NumPy float32 restatement of their formula; the old order stays selectable and is
    what the goldens' flow_direct_row_ltr holds."""
    import glob
    import os
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (70, 90)).astype(np.uint8)
    f = img.astype(np.float32)

    def numpy_rows(ksize, sigma, symm):
        k = oracle.gaussian_kernel(ksize, sigma)
        r = ksize // 2
        e = np.pad(f, ((0, 0), (r, r)), mode="reflect")
        S = lambda o: e[:, r + o:r + o + f.shape[1]]
        if symm and ksize == 3:
            return S(0) * k[1] + (S(-1) + S(1)) * k[2]
        if symm and ksize == 5:
            return S(0) * k[2] + (S(-1) + S(1)) * k[3] + (S(-2) + S(2)) * k[4]
        t = k[0] * S(-r)
        for i in range(1, ksize):
            t = t + k[i] * S(i - r)
        return t

    try:
        for ksize, sigma, differs in ((3, 0.0, False), (3, 0.5, True), (5, 0.9, True), (9, 1.5, False)):
            oracle.set_row_small_symm(True)
            a = oracle.gaussian_blur(f, ksize, sigma)
            oracle.set_row_small_symm(False)
            b = oracle.gaussian_blur(f, ksize, sigma)
            assert (not np.array_equal(a, b)) == differs, (ksize, sigma)
            if differs:
                assert np.abs(a - b).max() <= 2 ** -15 * 4        # an ulp or two of values < 256
            # the row pass itself against NumPy float32, both orders: feed a column-constant image so that the column pass
            # sees identical rows (k[r]*t + k[r+i]*(t + t) is then the same function of t for both settings)
            for symm in (True, False):
                oracle.set_row_small_symm(symm)
                rows = numpy_rows(ksize, sigma, symm).astype(np.float32)
                one = np.repeat(f[:1], 8, axis=0)
                got = oracle.gaussian_blur(one, ksize, sigma)
                k = oracle.gaussian_kernel(ksize, sigma)
                r = ksize // 2
                t = rows[:1]
                want = k[r] * t
                for i in range(1, r + 1):
                    want = want + k[r + i] * (t + t)
                np.testing.assert_array_equal(got[3:4], want.astype(np.float32))
        # whole pipeline: level 0 alone is exact either way; with a sigma = 0.5 level the flows differ by ~1e-6 px
        p, n, _ = translated_pair(96, 128, 31)
        oracle.set_row_small_symm(True)
        f0s, f1s = oracle.farneback(p, n, levels=0), oracle.farneback(p, n, levels=1)
        oracle.set_row_small_symm(False)
        f0l, f1l = oracle.farneback(p, n, levels=0), oracle.farneback(p, n, levels=1)
        np.testing.assert_array_equal(f0s, f0l)
        assert not np.array_equal(f1s, f1l) and np.abs(f1s - f1l).max() < 1e-4
        for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))):
            g = np.load(path, allow_pickle=False)
            kw = dict(eval(str(g["params"])))
            np.testing.assert_array_equal(oracle.farneback(g["prev"], g["next"], box_mode=oracle.BOX_BLOCKED, **kw),
                                          g["flow_direct_row_ltr"])
    finally:
        oracle.set_row_small_symm(True)


def test_rejects_bad_arguments(oracle):
    p, n, _ = translated_pair(64, 64, 1)
    with pytest.raises(ValueError):
        oracle.farneback(p, n, pyr_scale=1.0)
    with pytest.raises(ValueError):
        oracle.farneback(p, n, flags=4)       # OPTFLOW_USE_INITIAL_FLOW is not restated
    with pytest.raises(ValueError):
        oracle.farneback(p, n, winsize=1)


@pytest.mark.parametrize("zoom,angle,shift", [(1.03, 0.8, (1.6, -0.7)), (0.97, -2.0, (-2.3, 1.1)), (1.06, 0.0, (0.0, 0.0))])
def test_warped_family_ground_truth(oracle, zoom, angle, shift):
    """The FPV-like input family (synth.warped_pair: zoom about a focus + rotation + sub-pixel shift + an occluding patch that
    moves on its own): the oracle's flow follows the analytic ground truth away from the borders and the patch, the radial field
    really is non-uniform, and the three box-filter summation orders still agree to 1e-4 px on it."""
    from hackathonopticalflow_amd.synth import warped_pair
    h, w = 270, 480
    a, b, gt, valid = warped_pair(h, w, 77, zoom=zoom, angle_deg=angle, shift=shift)
    assert a.dtype == np.uint8 and a.shape == (h, w) and gt.shape == (h, w, 2)
    flow = oracle.farneback(a, b, levels=3)
    inner = valid.copy()
    inner[:34] = inner[-34:] = False
    inner[:, :34] = inner[:, -34:] = False
    e = np.linalg.norm(flow - gt, axis=-1)
    assert e[inner].mean() < 0.35, e[inner].mean()
    if zoom != 1.0:
        spread = np.linalg.norm(gt[inner] - gt[inner].mean(0), axis=-1).max()
        assert spread > 3.0                                     # several pixels of variation across the frame
    d = np.linalg.norm(flow - oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_BLOCKED), axis=-1)
    assert d.max() < 1e-4
