"""CPU checks of oracle/frontend_oracle.c (BGR->gray, INTER_AREA, HSV->BGR) and of the
USE_INITIAL_FLOW branch of the Farneback oracle.  PARITY UNPINNED (no cv2, no reference fixtures):
what pins these restatements is closed forms and independent float64 restatements."""
import colorsys

import numpy as np
import pytest

from hackathonopticalflow_amd.synth import translated_pair


def test_gray_coefficients(oracle):
    import ctypes as C
    for variant, shift in ((oracle.GRAY_15BIT, 15), (oracle.GRAY_14BIT, 14)):
        cb, cg, cr, sh = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        oracle.lib().ofo_gray_coeffs(variant, C.byref(cb), C.byref(cg), C.byref(cr), C.byref(sh))
        assert sh.value == shift
        assert cb.value + cg.value + cr.value == 1 << shift          # CV_Assert in RGB2Gray<uchar>
        for c, f in ((cr, 0.299), (cg, 0.587), (cb, 0.114)):
            assert abs(c.value / (1 << shift) - f) < 1e-4


@pytest.mark.parametrize("variant", [0, 1])
def test_bgr2gray_properties(oracle, variant):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    g = oracle.bgr2gray(img, variant)
    ref = img[..., 0] * 0.114 + img[..., 1] * 0.587 + img[..., 2] * 0.299
    assert np.abs(g.astype(np.float64) - ref).max() <= 0.51 + 1e-3 * 255
    # a gray pixel stays what it is (the coefficients sum to one)
    v = np.arange(256, dtype=np.uint8)
    grey = np.stack([v, v, v], -1)[None]
    np.testing.assert_array_equal(oracle.bgr2gray(grey, variant)[0], v)
    # pure channels at 255: round(255 * c)
    prim = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    np.testing.assert_array_equal(oracle.bgr2gray(prim, variant)[0], [29, 150, 76])


def _area_f64(src, dw, dh):
    """Independent float64 area resampling: exact overlap of each destination cell with the source pixels."""
    sh, sw = src.shape[:2]

    def weights(ss, ds):
        sc = ss / ds
        W = np.zeros((ds, ss))
        for d in range(ds):
            a, b = d * sc, min((d + 1) * sc, ss)
            for s_ in range(int(np.floor(a)), int(np.ceil(b))):
                W[d, s_] = max(0.0, min(b, s_ + 1) - max(a, s_))
            W[d] /= W[d].sum()
        return W
    Wx, Wy = weights(sw, dw), weights(sh, dh)
    return np.einsum("ab,bcd,ec->aed", Wy, src.astype(np.float64), Wx)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1920 // 4, 1080 // 4, 15, 9), (250, 130, 8, 5), (97, 83, 13, 11), (64, 48, 64, 48),
                                         (256, 128, 8, 4), (90, 60, 30, 20), (64, 32, 32, 16), (77, 33, 11, 33)])
def test_resize_area_vs_float64(oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(8)
    src = rng.standard_normal((sh, sw, 2)).astype(np.float32)
    got = oracle.resize_area(src, dw, dh)
    ref = _area_f64(src, dw, dh)
    assert got.shape == (dh, dw, 2)
    assert np.abs(got - ref).max() < 2e-6 * max(1.0, np.abs(ref).max()) + 1e-6
    # a constant field stays constant
    c = np.full((sh, sw, 2), 3.25, np.float32)
    assert np.abs(oracle.resize_area(c, dw, dh) - 3.25).max() < 1e-5


def test_resize_area_integer_factor_is_block_mean(oracle):
    rng = np.random.default_rng(9)
    src = rng.integers(-8, 9, (48, 64, 2)).astype(np.float32)       # small integers: float sums are exact
    got = oracle.resize_area(src, 8, 6)
    ref = src.reshape(6, 8, 8, 8, 2).mean(axis=(1, 3))
    np.testing.assert_array_equal(got, ref.astype(np.float32))


@pytest.mark.parametrize("ss,ds", [(1080, 34), (1920, 60), (135, 34), (100, 7), (33, 32)])
def test_area_table_rows_sum_to_one(oracle, ss, ds):
    si, di, al = oracle.area_tab(ss, ds)
    assert di.min() == 0 and di.max() == ds - 1 and (np.diff(di) >= 0).all()
    assert si.min() >= 0 and si.max() <= ss - 1
    sums = np.bincount(di, weights=al.astype(np.float64), minlength=ds)
    assert np.abs(sums - 1).max() < 1e-5


def test_hsv2bgr_against_colorsys(oracle):
    H, S, V = np.meshgrid(np.arange(0, 181), [0, 1, 64, 128, 200, 255], [0, 1, 50, 128, 254, 255], indexing="ij")
    hsv = np.stack([H, S, V], -1).astype(np.uint8).reshape(1, -1, 3)
    got = oracle.hsv2bgr_u8(hsv)[0].astype(int)
    ref = np.array([[round(c * 255) for c in colorsys.hsv_to_rgb((h % 180) / 180.0, s / 255.0, v / 255.0)][::-1]
                    for h, s, v in hsv[0].astype(float)])
    assert np.abs(got - ref).max() <= 1
    # S = 0 is gray; the six primaries/secondaries at full S, V
    np.testing.assert_array_equal(oracle.hsv2bgr_u8(np.array([[[77, 0, 123]]], np.uint8))[0, 0], [123, 123, 123])
    prim = np.array([[[h, 255, 255] for h in (0, 30, 60, 90, 120, 150)]], np.uint8)
    np.testing.assert_array_equal(oracle.hsv2bgr_u8(prim)[0], [[0, 0, 255], [0, 255, 255], [0, 255, 0], [255, 255, 0],
                                                                [255, 0, 0], [255, 0, 255]])


def test_draw_hsv_numpy_known_directions(oracle):
    f = np.zeros((1, 4, 2), np.float32)
    f[0, 0] = (10, 0)      # right: ang = pi -> H = 90 (cyan), V = 40
    f[0, 1] = (-10, 0)     # left: arctan2(0, -10) = pi -> ang = 2 pi -> H = 180 (wraps to red)
    f[0, 2] = (0, 100)     # down: ang = 3pi/2 -> H = 135, V saturates at 255
    hsv = oracle.draw_hsv_planes_numpy(f)
    assert hsv[0, 0].tolist() == [90, 255, 40]
    assert hsv[0, 1, 0] in (179, 180) and hsv[0, 2].tolist() == [135, 255, 255]
    assert hsv[0, 3].tolist() == [90, 255, 0]          # zero flow: arctan2(0, 0) = 0
    assert oracle.draw_hsv_numpy(f)[0, 3].tolist() == [0, 0, 0]


def test_draw_flow_lines_numpy(oracle):
    rng = np.random.default_rng(3)
    f = (rng.standard_normal((100, 150, 2)) * 6).astype(np.float32)
    lines = oracle.draw_flow_lines_numpy((100, 150), f, 14)
    assert lines.dtype == np.int32 and lines.shape == (7 * 11, 2, 2)
    assert lines[0, 0].tolist() == [7, 7] and lines[1, 0].tolist() == [21, 7]     # x fastest
    x, y = lines[:, 0, 0], lines[:, 0, 1]
    exp = np.stack([x - f[y, x, 0].astype(np.float64), y - f[y, x, 1].astype(np.float64)], -1) + 0.5
    np.testing.assert_array_equal(lines[:, 1], np.trunc(exp).astype(np.int32))


# ----------------------------------------------------------------------------- OPTFLOW_USE_INITIAL_FLOW
def test_initial_flow_zero_equals_plain_call(oracle):
    a, b, _ = translated_pair(96, 128, 11, max_shift=3)
    ref = oracle.farneback(a, b, levels=2)
    got = oracle.farneback(a, b, levels=2, flags=4, init_flow=np.zeros((96, 128, 2), np.float32))
    np.testing.assert_array_equal(got, ref)


def test_initial_flow_helps_large_motion(oracle):
    # a shift larger than one scale can capture from zero: the true flow as a start must do better
    a, b, (tx, ty) = translated_pair(120, 160, 21, max_shift=3)
    h, w = a.shape
    base = np.random.default_rng(2).standard_normal((h + 64, w + 64))
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(base, 3.0)
    base = np.round((base - base.min()) * (255 / (base.max() - base.min()))).astype(np.uint8)
    a = base[32:32 + h, 32:32 + w]
    b = base[32 - 14:32 - 14 + h, 32 - 20:32 - 20 + w]                # flow = (20, 14)
    kw = dict(levels=0, winsize=15, iterations=2)
    plain = oracle.farneback(a, b, **kw)
    init = np.empty((h, w, 2), np.float32)
    init[...] = (20, 14)
    warm = oracle.farneback(a, b, flags=4, init_flow=init, **kw)
    inner = (slice(30, -30), slice(30, -30))
    e_plain = np.linalg.norm(plain[inner] - (20, 14), axis=-1).mean()
    e_warm = np.linalg.norm(warm[inner] - (20, 14), axis=-1).mean()
    assert e_warm < 0.2 and e_warm < e_plain


def test_initial_flow_coarsest_level_is_scaled_area_resize(oracle):
    a, b, _ = translated_pair(96, 128, 12, max_shift=2)
    init = (np.random.default_rng(4).standard_normal((96, 128, 2)) * 2).astype(np.float32)
    _, cap = oracle.farneback(a, b, levels=1, flags=4, init_flow=init, capture=True)
    exp = oracle.resize_area(init, 64, 48) * np.float32(0.5)
    np.testing.assert_array_equal(cap.flow_init[1], exp)
    with pytest.raises(ValueError):
        oracle.farneback(a, b, levels=1, flags=4)
    with pytest.raises(ValueError):
        oracle.farneback(a, b, flags=8)


def test_cv_circle_filled_shape_and_clipping(oracle):
    """cv2.circle's filled LINE_8 raster as restated: radius 6 has rows of half-width 6 5 5 5 4 3 0 (the one-pixel tips of small OpenCV
    discs), is symmetric, lies inside the Euclidean disc of radius r + 0.5 and clips at the image border."""
    img = np.zeros((15, 15), np.uint8)
    oracle.cv_circle_filled(img, (7, 7), 6, 1)
    half = [int(img[7 + d].sum() - 1) // 2 for d in range(7)]
    assert half == [6, 5, 5, 5, 4, 3, 0]
    np.testing.assert_array_equal(img, img[::-1])
    np.testing.assert_array_equal(img, img.T)
    for r in range(0, 32):
        big = np.zeros((2 * r + 5, 2 * r + 5), np.uint8)
        c = r + 2
        oracle.cv_circle_filled(big, (c, c), r, 1)
        yy, xx = np.nonzero(big)
        assert ((yy - c) ** 2 + (xx - c) ** 2 <= (r + 0.5) ** 2 + 1e-9).all()
        assert big[c, c - r] == 1 and big[c - r, c] == 1 and big.sum() >= 0.75 * np.pi * r * r
        # clipped: the same disc drawn partly outside equals the crop of the whole one
        crop = np.zeros((r + 3, r + 4), np.uint8)
        oracle.cv_circle_filled(crop, (1, 2), r, 1)
        np.testing.assert_array_equal(crop, big[c - 2:c - 2 + r + 3, c - 1:c - 1 + r + 4])


def test_draw_sparse_lamps_numpy_known_case(oracle):
    flow = np.array([[3, 4], [0, 0], [200, 0]], np.int32)        # |f| = 5 -> V = 60; 0 -> 50; 200 -> 255 (saturated)
    pts = np.array([[10, 10], [40, 10], [70, 10]], np.int32)
    bgr = oracle.draw_sparse_lamps_numpy(flow, pts, 90, 30)
    assert bgr[10, 10].tolist() == [0, 0, 60] and bgr[10, 16].tolist() == [0, 0, 60] and bgr[10, 17].tolist() == [0, 0, 0]
    assert bgr[10, 40].tolist() == [0, 0, 50] and bgr[4, 70].tolist() == [0, 0, 255] and bgr[3, 70].tolist() == [0, 0, 0]
    assert int((bgr[..., 2] > 0).sum()) == 3 * int((bgr[..., 2] == 60).sum())
    np.testing.assert_array_equal(oracle.cv_add_u8(np.full_like(bgr, 250), bgr)[10, 10], [250, 250, 255])


def test_cv_line8_raster_known_cases(oracle):
    """The restated cv2.line raster: end points included, 8-connected, one pixel per major-axis step, the same pixels whichever end is
    given first (leftToRight), and clipping that keeps only the visible part."""
    def pix(p1, p2, shape=(9, 12)):
        img = np.zeros(shape, np.uint8)
        oracle.cv_line8(img, p1, p2, 1)
        return sorted(zip(*np.nonzero(img)[::-1]))
    assert pix((1, 1), (6, 1)) == [(x, 1) for x in range(1, 7)]
    assert pix((2, 0), (2, 5)) == [(2, y) for y in range(6)]
    assert pix((0, 0), (5, 5)) == [(i, i) for i in range(6)]
    assert pix((3, 3), (3, 3)) == [(3, 3)]
    got = pix((0, 0), (10, 3))
    assert len(got) == 11 and got[0] == (0, 0) and got[-1] == (10, 3) and [x for x, _ in got] == list(range(11))
    assert all(abs(y - 0.3 * x) <= 0.5 + 1e-9 for x, y in got)
    for p1, p2 in (((0, 0), (10, 3)), ((1, 7), (9, 2)), ((4, 0), (6, 8)), ((11, 8), (0, 0))):
        assert pix(p1, p2) == pix(p2, p1)
    # clipped: a line through the image from outside to outside, one leaving it, one that misses it
    across = pix((-5, 4), (20, 4))
    assert across == [(x, 4) for x in range(12)]
    assert pix((3, 3), (3, 40)) == [(3, y) for y in range(3, 9)]
    assert pix((-3, -3), (-1, 20)) == [] and pix((14, 2), (30, 5)) == []
    ok, a, b = oracle.cv_clip_line(12, 9, (-4, -4), (30, 30))
    assert ok and a == (0, 0) and b == (8, 8)


def test_draw_flow_numpy_known_case(oracle):
    flow = np.zeros((28, 42, 2), np.float32)
    flow[7, 7] = (-5, 0)                 # drawn from (7, 7) to (12, 7)
    flow[21, 35] = (100, 0)              # leaves the image on the left: clipped at x = 0
    img = oracle.draw_flow_numpy((28, 42), flow, 14)
    assert img[..., 0].sum() == 0 and img[..., 2].sum() == 0 and set(np.unique(img[..., 1])) == {0, 255}
    g = img[..., 1] > 0
    assert g[7, 6:13].all() and not g[7, 13] and g[6, 7] and g[8, 7]            # the line, the radius-1 disc at its start
    assert g[21, 0:37].all() and not g[21, 37]
    assert g[7, 21] and g[6, 21] and g[7, 20] and g[7, 22] and not g[6, 20]      # zero flow: just the disc


def test_cv_circle_outline_radius_one(oracle):
    img = np.zeros((5, 5), np.uint8)
    oracle.cv_circle_filled(img, (2, 2), 1, 1, fill=False)
    assert sorted(zip(*np.nonzero(img))) == [(1, 2), (2, 1), (2, 3), (3, 2)]          # thickness 1: the four neighbours, not the centre
    big = np.zeros((21, 21), np.uint8)
    oracle.cv_circle_filled(big, (10, 10), 8, 1, fill=False)
    full = np.zeros((21, 21), np.uint8)
    oracle.cv_circle_filled(full, (10, 10), 8, 1)
    assert big.sum() < full.sum() and not (big & ~full).any() and big[10, 2] and big[2, 10] and not big[10, 10]
