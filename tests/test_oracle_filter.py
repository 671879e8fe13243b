"""The C twin of the reference's NumPy grid / vector filter / danger brightness
(oracle/filter_oracle.c) against the reference's own lines run by the real NumPy
(oracle/oracle.py: pathfinder_viewer.py:159-176, 204-217, 252-267 re-typed)."""
import numpy as np
import pytest


@pytest.mark.parametrize("w,h,step,P", [(1920, 1080, 30, 2304), (640, 480, 30, 352), (3840, 2160, 30, 9216),
                                        (1920, 1080, 14, None), (1000, 700, 30, None), (641, 479, 25, None),
                                        (97, 211, 30, None)])
def test_grid_points(oracle, w, h, step, P):
    ref = oracle.grid_points_numpy(w, h, step)
    got = oracle.grid_points_c(w, h, step)
    if P is not None:
        assert len(ref) == P
    np.testing.assert_array_equal(got, ref)


def test_grid_1080p_layout(oracle):
    pts = oracle.grid_points_numpy(1920, 1080, 30)
    assert pts.dtype == np.float32
    assert tuple(pts[0]) == (15, 15) and tuple(pts[1]) == (15, 45) and tuple(pts[36]) == (45, 15)
    assert tuple(pts[-1]) == (1905, 1065)


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("w,h", [(1920, 1080), (640, 480), (1000, 700)])
def test_vector_filter_mask_bit_exact(oracle, seed, w, h):
    rng = np.random.default_rng(seed)
    pts = oracle.grid_points_numpy(w, h, 30)
    scale = [0.01, 0.5, 3, 20][seed % 4]
    vec = (rng.standard_normal((len(pts), 2)) * scale).astype(np.float32)
    if seed == 5:
        vec[::7] = 0            # exact zeros and ties
        n = min(len(vec[1::7]), len(vec[2::7]))
        vec[1::7][:n] = vec[2::7][:n]
    mask, mod, iflow, _ = oracle.vector_filter_numpy(vec, pts, w, h)
    cm, cmod, ciflow, cv, thr = oracle.vector_filter_c(vec, pts, w, h)
    np.testing.assert_array_equal(cmod, mod)
    assert thr[0] == float(np.median(mod) * 1.0)
    assert thr[1] == float(np.percentile(mod, 99))
    np.testing.assert_array_equal(cm, mask)
    # integer flow and V: arctan2 / cos / sin correctly rounded on both sides (oracle._cr) -> equal
    np.testing.assert_array_equal(ciflow, iflow)
    v_ref = np.zeros(len(pts), np.uint8)
    v_ref[mask] = oracle.lamp_values_numpy(iflow[mask])
    np.testing.assert_array_equal(cv, v_ref)


@pytest.mark.parametrize("seed", range(4))
def test_denseof_filter_variant_mask_bit_exact(oracle, seed):
    """Older gate of DenseOF.py:228: modulus > median * 1.2 (float32 product), no percentile gate."""
    rng = np.random.default_rng(100 + seed)
    w, h = 1920, 1080
    pts = oracle.grid_points_numpy(w, h, 30)
    vec = (rng.standard_normal((len(pts), 2)) * [0.02, 1, 5, 40][seed]).astype(np.float32)
    mask, mod, iflow, _ = oracle.vector_filter_numpy(vec, pts, w, h, variant=1)
    cm, cmod, ciflow, _, _ = oracle.vector_filter_c(vec, pts, w, h, variant=1)
    np.testing.assert_array_equal(cmod, mod)
    np.testing.assert_array_equal(cm, mask)
    assert mask.sum() > 0 and not np.array_equal(mask, oracle.vector_filter_numpy(vec, pts, w, h, variant=0)[0])


def test_percentile_small_sizes(oracle):
    rng = np.random.default_rng(3)
    for P in (2, 3, 5, 16, 101, 352):
        pts = (rng.uniform(0, 300, (P, 2))).astype(np.float32)
        vec = rng.standard_normal((P, 2)).astype(np.float32)
        mask, mod, _, _ = oracle.vector_filter_numpy(vec, pts, 640, 480)
        cm, cmod, _, _, thr = oracle.vector_filter_c(vec, pts, 640, 480)
        assert thr[1] == float(np.percentile(mod, 99)) and thr[0] == float(np.median(mod))
        np.testing.assert_array_equal(cm, mask)


def test_danger_map_dense_adaptation(oracle):
    rng = np.random.default_rng(4)
    flow = rng.standard_normal((480, 640, 2)).astype(np.float32) * 4
    mask, v = oracle.danger_map_numpy(flow, 640, 480)
    assert mask.shape == (352,) and v.shape == (352,) and mask.dtype == np.uint8
    assert 0 < mask.sum() < 352 // 2 + 1
    assert np.all(v[mask == 0] == 0) and np.all(v[mask == 1] >= 50)


def test_correctly_rounded_trig_vs_this_numpy(oracle):
    """What the 'correctly rounded' contract costs against the literal NumPy lines on THIS machine: NumPy's float32
    arctan2 / cos / sin are SIMD approximations (measured here: arctan2 up to 3.2 ulp), so the integer vectors of
    pathfinder_viewer.py:169-171 can differ where a coordinate lands within an ulp of an integer + 0.5 boundary.
    Measured on 2 x 10^6 random vectors: 0.5 per 10^6 points; the bound asserted is 20 per 10^6."""
    rng = np.random.default_rng(0)
    n = 2_000_000
    vec = (rng.standard_normal((n, 2)) * 5).astype(np.float32)
    pts = np.stack([rng.integers(0, 1920, n), rng.integers(0, 1080, n)], 1).astype(np.float32)
    _, mod_a, if_cr, _ = oracle.vector_filter_numpy(vec, pts, 1920, 1080, cr=True)
    _, mod_b, if_np, _ = oracle.vector_filter_numpy(vec, pts, 1920, 1080, cr=False)
    np.testing.assert_array_equal(mod_a, mod_b)          # the moduli (and so the mask) do not depend on it
    bad = int((if_cr != if_np).any(axis=1).sum())
    assert bad <= 20 * n // 1_000_000, bad
    flow = vec.reshape(1000, 2000, 2)
    h_cr = oracle.draw_hsv_planes_numpy(flow, cr=True)[..., 0]
    h_np = oracle.draw_hsv_planes_numpy(flow, cr=False)[..., 0]
    assert int((h_cr != h_np).sum()) <= 50 * n // 1_000_000     # measured 3.5 per 10^6


def test_nan_vector_gives_empty_mask(oracle):
    """A NaN in the data makes np.median and np.percentile NaN, every comparison False: nothing is kept."""
    rng = np.random.default_rng(7)
    pts = oracle.grid_points_numpy(640, 480, 30)
    vec = rng.standard_normal((len(pts), 2)).astype(np.float32)
    vec[17, 0] = np.nan
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mask, _, _, _ = oracle.vector_filter_numpy(vec, pts, 640, 480)
    cm, _, _, cv, thr = oracle.vector_filter_c(vec, pts, 640, 480)
    assert not mask.any() and not cm.any() and not cv.any() and np.isnan(thr).all()
