"""GPU parity: the HIP path (through the C-ABI of include/ofarn.h) against the CPU oracle.

Bar (north_star): flow within a stated EPE tolerance of the OpenCV-algorithm oracle; danger-point
index sets bit-identical.  What is asserted here:
  * every stage, and the whole pipeline, BIT-EXACT against the oracle run with the block-restarted
    running-sum box order (oracle BOX_BLOCKED: same IEEE operations in the same order as the kernels);
  * against the oracle in OpenCV's literal running-sum order: mean EPE <= 1e-5 px and
    max EPE <= 1e-3 px (TOL_* below) -- the only difference is the summation order of the box filter;
  * danger mask bit-exact against the reference's NumPy filter on the same flow; V, the integer vectors and the
    hue of draw_hsv bit-exact against the same NumPy lines with arctan2 / cos / sin evaluated correctly rounded
    (oracle._cr: NumPy's own float32 loops for those are CPU-dependent approximations, up to 3.2 ulp here), and within
    a MEASURED mismatch rate of the literal lines on this machine's NumPy (test_trig_outputs_vs_literal_numpy);
  * the danger index set from GPU flow against the one from the oracle's OpenCV-order flow: symmetric difference
    reported and bounded (test_danger_sets_gpu_flow_vs_opencv_order_flow);
  * inputs: integer translations AND the FPV-like warped family (zoom + rotation + sub-pixel shift + occluder).
"""
import ast
import glob
import os

import numpy as np
import pytest

from hackathonopticalflow_amd.synth import translated_pair, translated_pairs, warped_pair, warped_pairs

pytestmark = pytest.mark.gpu

TOL_MEAN_EPE = 1e-5   # px, GPU vs oracle in OpenCV's running-sum order
TOL_MAX_EPE = 1e-3    # px

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.fixture(scope="module")
def H():
    import hackathonopticalflow_amd as H
    H.load_library()
    return H


@pytest.fixture(params=["march", "tile"])
def iter_kernel(request, monkeypatch):
    """The fused iteration exists twice with identical results: the marching kernel (throughput; chosen when its grid fills
    the chip) and the tile kernel (latency; chosen for small grids).  Left alone, a single small pair would only ever reach
    the tile kernel, so the pipeline tests force each in turn (OFARN_TILE, read when a context is created)."""
    import hackathonopticalflow_amd as HH
    monkeypatch.setenv("OFARN_TILE", "0" if request.param == "march" else "1")
    HH.close_cached_engines()          # the switch is read when a context is created: drop the drop-in's cached ones
    yield request.param
    HH.close_cached_engines()


def epe(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64), axis=-1)


# ------------------------------------------------------------------------------------ stages
@pytest.mark.parametrize("w,h,levels", [(640, 480, 3), (333, 251, 2), (1920, 1080, 5)])
def test_stage_level_image_bit_exact(H, oracle, w, h, levels):
    img, _, _ = translated_pair(h, w, 7)
    with H.FarnebackEngine(w, h, 1, levels=levels) as eng:
        for k, (lw, lh, ks, sg) in enumerate(H.level_plan(w, h, levels=levels)):
            assert (lw, lh, sg, ks) == oracle.level_geom(w, h, 0.5, k)
            got = eng.stage_level_image(img, k)
            ref = oracle.level_image(img, ks, sg, lw, lh)
            np.testing.assert_array_equal(got, ref, err_msg=f"level {k}")


@pytest.mark.parametrize("w,h,n,sigma", [(320, 240, 5, 1.2), (97, 83, 7, 1.5), (64, 33, 3, 0.0), (200, 40, 5, 1.1)])
def test_stage_polyexp_bit_exact(H, oracle, w, h, n, sigma):
    rng = np.random.default_rng(1)
    I = rng.uniform(0, 255, (h, w)).astype(np.float32)
    with H.FarnebackEngine(w, h, 1, poly_n=n, poly_sigma=sigma) as eng:
        got = eng.stage_polyexp(I)
    np.testing.assert_array_equal(got, oracle.polyexp(I, n, sigma))


@pytest.mark.parametrize("w,h", [(320, 240), (97, 83), (33, 40)])
def test_stage_update_matrices_bit_exact(H, oracle, w, h):
    rng = np.random.default_rng(2)
    R0 = rng.standard_normal((h, w, 5)).astype(np.float32)
    R1 = rng.standard_normal((h, w, 5)).astype(np.float32)
    flow = (rng.standard_normal((h, w, 2)) * 4).astype(np.float32)
    flow[0, :5] = -9          # out-of-bounds branch
    flow[-1, -5:] = 9
    flow[5, 5] = (0.0, 0.0)
    with H.FarnebackEngine(w, h, 1) as eng:
        got = eng.stage_update_matrices(R0, R1, flow)
    np.testing.assert_array_equal(got, oracle.update_matrices(R0, R1, flow))


@pytest.mark.parametrize("w,h,winsize", [(320, 240, 15), (97, 83, 9), (70, 50, 8), (40, 33, 3), (64, 48, 41),
                                         (300, 200, 127), (33, 100, 2)])
def test_stage_blur_solve_bit_exact(H, oracle, w, h, winsize):
    rng = np.random.default_rng(3)
    M = (rng.standard_normal((h, w, 5)) * 10).astype(np.float32)
    z5, z2 = np.zeros((h, w, 5), np.float32), np.zeros((h, w, 2), np.float32)
    ref, _ = oracle.update_flow_blur(z5, z5, z2, M, winsize, False, oracle.BOX_BLOCKED)
    with H.FarnebackEngine(w, h, 1, winsize=winsize) as eng:
        got = eng.stage_blur_solve(M)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("w,h,winsize", [(97, 83, 15), (200, 150, 2), (64, 130, 3), (333, 70, 10), (70, 333, 21), (150, 140, 63), (260, 200, 127),
                                         (31, 5, 15), (5, 31, 15), (1, 1, 15), (65, 33, 4)])
def test_stage_blur_solve_literal_order_bit_exact(H, oracle, w, h, winsize):
    """FarnebackUpdateFlow_Blur alone in optflowgf.cpp's literal order (k_vsum_running + k_hsum_running_solve) on random matrices:
    every window size class (even, odd, up to 127), images smaller than the window, tiles that do not divide the image."""
    rng = np.random.default_rng(winsize * 1000 + w)
    M = (rng.standard_normal((h, w, 5)) * 10).astype(np.float32)
    z5, z2 = np.zeros((h, w, 5), np.float32), np.zeros((h, w, 2), np.float32)
    ref, _ = oracle.update_flow_blur(z5, z5, z2, M, winsize, False, oracle.BOX_RUNNING)
    with H.FarnebackEngine(w, h, 1, winsize=winsize) as eng:
        eng.set_option("box_order", 1)
        np.testing.assert_array_equal(eng.stage_blur_solve(M), ref)


@pytest.mark.parametrize("sw,sh,dw,dh", [(160, 120, 320, 240), (60, 34, 120, 68), (120, 68, 240, 135), (49, 42, 97, 83)])
def test_stage_flow_upsample_bit_exact(H, oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(4)
    f = (rng.standard_normal((sh, sw, 2)) * 3).astype(np.float32)
    with H.FarnebackEngine(dw, dh, 1) as eng:
        got = eng.stage_flow_upsample(f, dw, dh)
    np.testing.assert_array_equal(got, oracle.resize_linear(f, dw, dh) * np.float32(2))


# ------------------------------------------------------------------------------------ pipeline
CASES = [
    # (w, h, seed, kwargs)   -- first row is BASELINE config 1 (640x480, DenseOF.py defaults)
    (640, 480, 1001, {}),
    (333, 251, 11, dict(levels=2)),
    (200, 160, 12, dict(levels=0)),
    (256, 192, 13, dict(levels=4, pyr_scale=0.8, winsize=8, iterations=2)),
    (180, 130, 14, dict(levels=1, poly_n=7, poly_sigma=1.5, winsize=21, iterations=1)),
    (64, 64, 15, dict(levels=5)),          # cropped by min_size to 1 reduction
    # window sizes with a fused-kernel instantiation (m = winsize/2 in {3, 5, 10}) and multi-strip heights
    (300, 420, 16, dict(levels=2, winsize=7)),
    (521, 333, 17, dict(levels=2, winsize=11, iterations=2)),
    (300, 260, 18, dict(levels=1, winsize=21)),
    (300, 260, 19, dict(levels=1, winsize=20)),
    (310, 270, 20, dict(levels=1, winsize=9)),             # m = 4, 6, 8, 9: the remaining fused instantiations
    (290, 250, 21, dict(levels=2, winsize=13, iterations=2)),
    (330, 240, 22, dict(levels=1, winsize=17)),
    (300, 300, 23, dict(levels=1, winsize=19, iterations=2)),
    (300, 260, 24, dict(levels=1, winsize=23)),            # m = 11: generic kernels
    (640, 480, 25, dict(levels=3, poly_n=7, poly_sigma=1.5)),   # the marching polynomial expansion with radius 7
    (200, 150, 26, dict(levels=1, poly_n=3, poly_sigma=1.0)),   # radius 3: generic polynomial expansion
]


@pytest.mark.parametrize("w,h,seed,kw", CASES)
def test_pipeline_vs_oracle(H, oracle, iter_kernel, w, h, seed, kw):
    prev, nxt, _ = translated_pair(h, w, seed, max_shift=5)
    got = H.calculate_optical_flow(prev, nxt, **kw)
    assert got.shape == (h, w, 2) and got.dtype == np.float32
    np.testing.assert_array_equal(got, oracle.farneback(prev, nxt, box_mode=oracle.BOX_BLOCKED, **kw))
    e = epe(got, oracle.farneback(prev, nxt, box_mode=oracle.BOX_RUNNING, **kw))
    if kw.get("winsize", 15) >= 15:
        assert e.mean() <= TOL_MEAN_EPE and e.max() <= TOL_MAX_EPE, (e.mean(), e.max())
    else:
        # Small windows leave isolated pixels whose 2x2 system is nearly singular (only the +1e-3 keeps it
        # solvable); there OpenCV's own float-rounded running row differences move the result by up to
        # ~0.1 px between ANY two summation orders (CPU oracle RUNNING vs DIRECT shows the same).  The
        # bulk still agrees to 1e-5.
        assert e.mean() <= 1e-4 and np.quantile(e, 0.999) <= 1e-3 and e.max() <= 0.5, (e.mean(), e.max())


def test_stage_kernels_generic_variants(H, oracle, monkeypatch):
    """The stage tests above run the specialised kernels where they exist (marching poly
    expansion, LDS-staged row pass); this repeats two of them on the generic kernels."""
    monkeypatch.setenv("OFARN_FORCE_GENERIC", "1")
    rng = np.random.default_rng(1)
    I = rng.uniform(0, 255, (83, 300)).astype(np.float32)
    with H.FarnebackEngine(300, 83, 1) as eng:
        np.testing.assert_array_equal(eng.stage_polyexp(I), oracle.polyexp(I, 5, 1.2))
    img, _, _ = translated_pair(251, 333, 7)
    with H.FarnebackEngine(333, 251, 1, levels=2) as eng:
        for k, (lw, lh, ks, sg) in enumerate(H.level_plan(333, 251, levels=2)):
            np.testing.assert_array_equal(eng.stage_level_image(img, k), oracle.level_image(img, ks, sg, lw, lh))


def test_generic_and_fused_paths_agree(H, oracle, monkeypatch):
    """winsize 15 dispatches to the fused marching kernel; OFARN_FORCE_GENERIC=1 keeps the unfused
    generic kernels.  Both must equal the oracle bit for bit (multi-strip, multi-block sizes)."""
    prev, nxt, _ = translated_pair(300, 520, 41, max_shift=6)
    ref = oracle.farneback(prev, nxt, levels=2, box_mode=oracle.BOX_BLOCKED)
    with H.FarnebackEngine(520, 300, 1, levels=2) as eng:
        np.testing.assert_array_equal(eng.calc(prev, nxt), ref)
    monkeypatch.setenv("OFARN_FORCE_GENERIC", "1")
    with H.FarnebackEngine(520, 300, 1, levels=2) as eng:
        np.testing.assert_array_equal(eng.calc(prev, nxt), ref)
    monkeypatch.delenv("OFARN_FORCE_GENERIC")
    for iters in (1, 2, 4):
        ref = oracle.farneback(prev, nxt, levels=1, iterations=iters, box_mode=oracle.BOX_BLOCKED)
        np.testing.assert_array_equal(H.calculate_optical_flow(prev, nxt, levels=1, iterations=iters), ref)


@pytest.mark.parametrize("w,h", [(9, 7), (33, 17), (16, 64), (257, 3), (5, 300), (1, 1), (40, 1), (1, 40)])
def test_degenerate_sizes(H, oracle, iter_kernel, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    prev = rng.integers(0, 256, (h, w)).astype(np.uint8)
    nxt = rng.integers(0, 256, (h, w)).astype(np.uint8)
    for kw in (dict(levels=3), dict(levels=0, winsize=5, iterations=2)):
        got = H.calculate_optical_flow(prev, nxt, **kw)
        np.testing.assert_array_equal(got, oracle.farneback(prev, nxt, box_mode=oracle.BOX_BLOCKED, **kw))


def test_pipeline_1080p_L5_config2(H, oracle, iter_kernel):
    """BASELINE config 2: one 1920x1080 pair, levels=5, iterations=3, seed 2001."""
    prev, nxt, (tx, ty) = translated_pair(1080, 1920, 2001)
    got = H.calculate_optical_flow(prev, nxt, levels=5)
    ref = oracle.farneback(prev, nxt, levels=5, box_mode=oracle.BOX_BLOCKED)
    np.testing.assert_array_equal(got, ref)
    e = epe(got, oracle.farneback(prev, nxt, levels=5))
    assert e.mean() <= TOL_MEAN_EPE and e.max() <= TOL_MAX_EPE, (e.mean(), e.max())
    gt = epe(got[32:-32, 32:-32], np.float32([tx, ty])[None, None])
    assert gt.mean() < 0.1


def test_pipeline_4k_L6_I5_config5(H, oracle):
    """BASELINE config 5 shape: 3840x2160, levels=6 (7 scales, top 60x34, 159-tap level blur), iterations=5."""
    prev, nxt, (tx, ty) = translated_pair(2160, 3840, 5001)
    got = H.calculate_optical_flow(prev, nxt, levels=6, iterations=5)
    ref = oracle.farneback(prev, nxt, levels=6, iterations=5, box_mode=oracle.BOX_BLOCKED)
    np.testing.assert_array_equal(got, ref)
    gt = epe(got[64:-64, 64:-64], np.float32([tx, ty])[None, None])
    assert gt.mean() < 0.1


@pytest.mark.parametrize("w,h,levels,iterations", [(7680, 4320, 7, 3), (4097, 2161, 6, 3), (8191, 33, 3, 2), (33, 8191, 3, 2)])
def test_beyond_baseline_sizes(H, oracle, w, h, levels, iterations):
    """Past BASELINE's largest shape: an 8K pair with 8 scales (rows of 7680 no longer fit the row-pass kernels' LDS staging at the
    coarse levels' tap counts), odd 4K-plus sizes (no level is an exact half), frames of 33 rows or columns (cropped to one scale).
    Bit-exact against the oracle in the device's summation order, like every other size."""
    a, b, (tx, ty) = translated_pair(h, w, 5, max_shift=6)
    with H.FarnebackEngine(w, h, 1, levels=levels, iterations=iterations) as eng:
        got = eng.calc(a, b)
    np.testing.assert_array_equal(got, oracle.farneback(a, b, levels=levels, iterations=iterations, box_mode=oracle.BOX_BLOCKED))
    if min(w, h) > 1000:
        assert epe(got[64:-64, 64:-64], np.float32([tx, ty])[None, None]).mean() < 0.1


def test_full_size_properties_config3(H):
    """Size-independent properties at BASELINE's full size (batch of 1080p pairs, levels=5):
    a pair and its duplicate in the same batch give identical flow and danger maps; the interior flow
    of every pair is its ground-truth translation; the wave split does not change results."""
    frames, shifts = translated_pairs(3, 1080, 1920, 3000)
    batch = np.concatenate([frames, frames[:2], frames[2:4]])   # 5 pairs: 0 1 2 0 1
    with H.FarnebackEngine(1920, 1080, 2, levels=5) as eng:
        flow, mask, v = eng.calc_batch(batch, H.PAIRS_INDEPENDENT)
    np.testing.assert_array_equal(flow[0], flow[3])
    np.testing.assert_array_equal(mask[0], mask[3])
    np.testing.assert_array_equal(v[0], v[3])
    np.testing.assert_array_equal(flow[1], flow[4])
    for i in range(3):
        gt = epe(flow[i][32:-32, 32:-32], shifts[i].astype(np.float32)[None, None])
        assert gt.mean() < 0.1, (i, shifts[i], gt.mean())
    with H.FarnebackEngine(1920, 1080, 5, levels=5) as eng:
        flow2, mask2, _ = eng.calc_batch(batch, H.PAIRS_INDEPENDENT)
    np.testing.assert_array_equal(flow, flow2)
    np.testing.assert_array_equal(mask, mask2)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(H, iter_kernel, path):
    g = np.load(path, allow_pickle=False)
    kw = dict(ast.literal_eval(str(g["params"])))
    got = H.calculate_optical_flow(g["prev"], g["next"], **kw)
    np.testing.assert_array_equal(got, g["flow_direct"])
    e = epe(got, g["flow_running"])
    assert e.mean() <= TOL_MEAN_EPE and e.max() <= TOL_MAX_EPE
    mask, v = H.danger_map(got, 30)
    np.testing.assert_array_equal(mask, g["mask"])
    np.testing.assert_array_equal(v, g["v"])


def test_row_order_of_small_gaussian_kernels(H, oracle, monkeypatch):
    """GaussianBlur's row pass of a 3- or 5-tap kernel runs in SymmRowSmallFilter's order, S[0]*k0 + (S[-1]+S[1])*k1 (+ ...), on
    both sides by default (oracle OFO_ROW_SMALL_SYMM; every row-pass kernel: generic, LDS, multi-level, direct 1/2).  Level
    images with ksize 3 (pyr_scale 0.5, level 1) and ksize 5 (pyr_scale 0.6, level 2) are bit-exact on every path, the plain
    left-to-right order of rounds 1-2 is still selectable on both sides (OFARN_ROW_LTR=1 when the context is created /
    oracle.set_row_small_symm(False)) and reproduces the goldens' flow_direct_row_ltr."""
    img, _, _ = translated_pair(240, 320, 9)
    for env in ({}, {"OFARN_FORCE_GENERIC": "1"}, {"OFARN_DIRECT_MIN_FRAMES": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        for ps, lv in ((0.5, 2), (0.6, 3)):
            with H.FarnebackEngine(320, 240, 1, levels=lv, pyr_scale=ps) as eng:
                ks_seen = set()
                for k, (lw, lh, ks, sg) in enumerate(H.level_plan(320, 240, levels=lv, pyr_scale=ps)):
                    ks_seen.add(ks)
                    np.testing.assert_array_equal(eng.stage_level_image(img, k), oracle.level_image(img, ks, sg, lw, lh),
                                                  err_msg=f"{env} pyr_scale {ps} level {k} ksize {ks}")
                assert 3 in ks_seen and (ps == 0.5 or 5 in ks_seen)
            a, b, _ = translated_pair(240, 320, 10, max_shift=3)
            with H.FarnebackEngine(320, 240, 1, levels=lv, pyr_scale=ps) as eng:
                np.testing.assert_array_equal(eng.calc(a, b), oracle.farneback(a, b, levels=lv, pyr_scale=ps, box_mode=oracle.BOX_BLOCKED))
        for k_ in env:
            monkeypatch.delenv(k_)
    monkeypatch.setenv("OFARN_ROW_LTR", "1")
    oracle.set_row_small_symm(False)
    try:
        for path in GOLDEN:
            g = np.load(path, allow_pickle=False)
            kw = dict(ast.literal_eval(str(g["params"])))
            hh, ww = g["prev"].shape
            with H.FarnebackEngine(ww, hh, 1, **kw) as eng:
                got = eng.calc(g["prev"], g["next"])
            np.testing.assert_array_equal(got, g["flow_direct_row_ltr"])
            assert not np.array_equal(got, g["flow_direct"]) or kw.get("levels", 3) == 0
        with H.FarnebackEngine(320, 240, 1, levels=2) as eng:
            for k, (lw, lh, ks, sg) in enumerate(H.level_plan(320, 240, levels=2)):
                np.testing.assert_array_equal(eng.stage_level_image(img, k), oracle.level_image(img, ks, sg, lw, lh))
    finally:
        oracle.set_row_small_symm(True)


@pytest.mark.parametrize("w,h,kw", [(320, 240, dict(levels=2)), (333, 251, dict(levels=1, winsize=9, iterations=2)),
                                    (200, 160, dict(levels=0, winsize=20)),
                                    # the fused Gaussian iteration kernel: every instantiation (m = 3..8), multi-strip heights,
                                    # one and several iterations (first iteration = on-the-fly upsample)
                                    (521, 333, dict(levels=2, winsize=7)), (300, 420, dict(levels=2, winsize=11, iterations=2)),
                                    (640, 480, dict(levels=3, winsize=13)), (300, 420, dict(levels=1, winsize=17, iterations=4)),
                                    (1920, 1080, dict(levels=5)), (97, 83, dict(levels=1, winsize=19))])
def test_farneback_gaussian_flag(H, oracle, w, h, kw):
    """flags=OPTFLOW_FARNEBACK_GAUSSIAN (256): FarnebackUpdateFlow_GaussianBlur, float32 separable window.  winsize 6..17
    runs the fused marching kernel (kernels_gauss.hip), wider windows the unfused pair; both against the same oracle."""
    prev, nxt, _ = translated_pair(h, w, 51, max_shift=5)
    got = H.calculate_optical_flow(prev, nxt, flags=256, **kw)
    np.testing.assert_array_equal(got, oracle.farneback(prev, nxt, flags=256, **kw))
    rng = np.random.default_rng(5)
    M = (rng.standard_normal((h, w, 5)) * 10).astype(np.float32)
    z5, z2 = np.zeros_like(M), np.zeros((h, w, 2), np.float32)
    ws = kw.get("winsize", 15)
    ref, _ = oracle.update_flow_gaussian(z5, z5, z2, M, ws, False)
    with H.FarnebackEngine(w, h, 1, winsize=ws, flags=256) as eng:
        np.testing.assert_array_equal(eng.stage_blur_solve(M), ref)


def test_constant_and_flow_reuse(H):
    img = np.full((96, 128), 200, np.uint8)
    out = np.full((96, 128, 2), 7, np.float32)
    ret = H.calculate_optical_flow(img, img, out, levels=1)
    assert ret is out and np.all(out == 0)
    # cv2 positional order
    ret2 = H.calcOpticalFlowFarneback(img, img, None, 0.5, 1, 15, 3, 5, 1.2, 0)
    assert np.all(ret2 == 0)


def test_strided_input(H, oracle):
    big_p, big_n, _ = translated_pair(200, 300, 31)
    prev, nxt = big_p[10:170, 20:260], big_n[10:170, 20:260]    # non-contiguous views
    got = H.calculate_optical_flow(prev, nxt, levels=2)
    ref = oracle.farneback(np.ascontiguousarray(prev), np.ascontiguousarray(nxt), levels=2, box_mode=oracle.BOX_BLOCKED)
    np.testing.assert_array_equal(got, ref)


# ------------------------------------------------------------------------------------ batch + danger map
def test_batch_modes_and_waves(H, oracle, iter_kernel):
    h, w, n_pairs = 120, 160, 5
    frames, _ = translated_pairs(n_pairs, h, w, 3000, max_shift=4)
    kw = dict(levels=2)
    ref = np.stack([oracle.farneback(frames[2 * i], frames[2 * i + 1], box_mode=oracle.BOX_BLOCKED, **kw)
                    for i in range(n_pairs)])
    for wave in (1, 2, 8):
        with H.FarnebackEngine(w, h, wave, **kw) as eng:
            flow, mask, v = eng.calc_batch(frames, H.PAIRS_INDEPENDENT)
            np.testing.assert_array_equal(flow, ref)
            for i in range(n_pairs):
                m_ref, v_ref = oracle.danger_map_numpy(flow[i], w, h, 30)
                np.testing.assert_array_equal(mask[i], m_ref)
                np.testing.assert_array_equal(v[i], v_ref)
            # video order: pair i = frames (i, i+1)
            flow_c, _, _ = eng.calc_batch(frames[:6], H.PAIRS_CONSECUTIVE, want_danger=False)
            assert flow_c.shape[0] == 5
            np.testing.assert_array_equal(flow_c[0], ref[0])
            np.testing.assert_array_equal(flow_c[2], ref[1])
            mid = oracle.farneback(frames[1], frames[2], box_mode=oracle.BOX_BLOCKED, **kw)
            np.testing.assert_array_equal(flow_c[1], mid)


def test_batch_empty_and_ragged(H):
    with H.FarnebackEngine(160, 120, 4, levels=1) as eng:
        flow, mask, v = eng.calc_batch(np.zeros((0, 120, 160), np.uint8))
        assert flow.shape == (0, 120, 160, 2) and mask.shape[0] == 0
        flow, _, _ = eng.calc_batch(np.zeros((1, 120, 160), np.uint8), H.PAIRS_CONSECUTIVE)
        assert flow.shape[0] == 0
        with pytest.raises(ValueError):
            eng.calc_batch(np.zeros((3, 120, 160), np.uint8), H.PAIRS_INDEPENDENT)   # odd frame count
        with pytest.raises(ValueError):
            eng.calc_batch(np.zeros((2, 500, 500), np.uint8))                          # larger than ctx


@pytest.mark.parametrize("w,h,step", [(1920, 1080, 30), (640, 480, 30), (1000, 700, 14)])
def test_danger_map_mask_bit_exact_random_flow(H, oracle, w, h, step):
    rng = np.random.default_rng(w)
    flows = (rng.standard_normal((3, h, w, 2)) * np.float32([0.05, 2, 30])[:, None, None, None]).astype(np.float32)
    flows[1, ::2] = 0     # ties / exact zeros at many grid points
    with H.FarnebackEngine(w, h, 1, grid_step=step) as eng:
        mask, v = eng.danger_map(flows)
    for i in range(3):
        m_ref, v_ref = oracle.danger_map_numpy(flows[i], w, h, step)
        np.testing.assert_array_equal(mask[i], m_ref)
        np.testing.assert_array_equal(v[i], v_ref)


def test_danger_map_variants_and_integer_flow(H, oracle):
    rng = np.random.default_rng(9)
    w, h = 640, 480
    flow = (rng.standard_normal((h, w, 2)) * 3).astype(np.float32)
    for variant in (H.FILTER_VIEWER, H.FILTER_DENSEOF):
        mask, v, iflow = H.danger_map(flow, 30, filter_variant=variant, return_flow=True)
        m_ref, v_ref, if_ref = oracle.danger_map_numpy(flow, w, h, 30, variant=variant, return_flow=True)
        np.testing.assert_array_equal(mask, m_ref)
        assert iflow.shape == (len(m_ref), 2) and iflow.dtype == np.int32
        np.testing.assert_array_equal(iflow, if_ref)
        np.testing.assert_array_equal(v, v_ref)


def test_device_resident_batch_torch(H, oracle):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    h, w, n_pairs = 120, 160, 6
    frames, _ = translated_pairs(n_pairs, h, w, 4000, max_shift=4)
    P = len(H.grid_points(w, h, 30))
    d_frames = torch.from_numpy(frames).cuda()
    d_flow = torch.empty((n_pairs, h, w, 2), dtype=torch.float32, device="cuda")
    d_mask = torch.zeros((n_pairs, P), dtype=torch.uint8, device="cuda")
    d_v = torch.zeros((n_pairs, P), dtype=torch.uint8, device="cuda")
    with H.FarnebackEngine(w, h, 4, levels=2) as eng:
        st = torch.cuda.current_stream().cuda_stream
        eng.calc_batch_device(d_frames, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, d_mask, d_v, stream=st)
        torch.cuda.synchronize()
        flow = d_flow.cpu().numpy()
        for i in range(n_pairs):
            ref = oracle.farneback(frames[2 * i], frames[2 * i + 1], levels=2, box_mode=oracle.BOX_BLOCKED)
            np.testing.assert_array_equal(flow[i], ref)
            m_ref, _ = oracle.danger_map_numpy(flow[i], w, h, 30)
            np.testing.assert_array_equal(d_mask[i].cpu().numpy(), m_ref)
        # danger maps only (flow kept internal)
        d_mask2 = torch.zeros_like(d_mask)
        d_v2 = torch.zeros_like(d_v)
        eng.calc_batch_device(d_frames, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, None, d_mask2, d_v2, stream=st)
        torch.cuda.synchronize()
        assert torch.equal(d_mask, d_mask2) and torch.equal(d_v, d_v2)


def test_device_tensor_validation(H):
    torch = pytest.importorskip("torch")
    h, w = 64, 96
    fr = torch.zeros((2, h, w), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((1, h, w, 2), dtype=torch.float32, device="cuda")
    with H.FarnebackEngine(w, h, 1, levels=1) as eng:
        with pytest.raises(ValueError, match="uint8"):
            eng.calc_batch_device(fr.float(), 2, w, h, H.PAIRS_INDEPENDENT, fl, None, None)
        with pytest.raises(ValueError, match="at least"):
            eng.calc_batch_device(fr, 2, w, h, H.PAIRS_INDEPENDENT, fl[:, : h // 2], None, None)
        with pytest.raises(ValueError, match="contiguous"):
            eng.calc_batch_device(fr, 2, w, h, H.PAIRS_INDEPENDENT, fl.permute(0, 2, 1, 3), None, None)
        with pytest.raises(ValueError, match="GPU"):
            eng.calc_batch_device(fr.cpu(), 2, w, h, H.PAIRS_INDEPENDENT, fl, None, None)
        with pytest.raises(ValueError, match="float32"):
            eng.flow_hsv_device(fl.double(), 1, w, h, None, torch.zeros((1, h, w, 3), dtype=torch.uint8, device="cuda"))


# ------------------------------------------------------------------------------------ errors
def test_argument_errors(H):
    a = np.zeros((64, 64), np.uint8)
    with pytest.raises(ValueError):
        H.calculate_optical_flow(a, np.zeros((64, 65), np.uint8))
    with pytest.raises(ValueError):
        H.calculate_optical_flow(a.astype(np.float32), a)
    with pytest.raises(ValueError):
        H.calculate_optical_flow(np.zeros((64, 64, 3), np.uint8), a)
    with pytest.raises(ValueError):
        H.calculate_optical_flow(a, a, pyr_scale=1.0)
    with pytest.raises(ValueError):
        H.calculate_optical_flow(a, a, winsize=1)
    with pytest.raises(NotImplementedError):
        H.calculate_optical_flow(a, a, flags=8)        # not a cv2 flag of this function
    with pytest.raises(ValueError):
        H.calculate_optical_flow(a, a, flags=4)        # OPTFLOW_USE_INITIAL_FLOW without `flow`


def test_direct_level_kernels_in_the_pipeline(H, oracle, monkeypatch):
    """k_level_direct (stage A of the 1/2, 1/4, 1/8 levels in one kernel) is chosen for waves of >= 16 frames of 1920 x 1080 (by pixel count); force it
    for single pairs, then switch it off for a small batch, and compare the whole pipeline with the oracle bit for bit."""
    monkeypatch.setenv("OFARN_DIRECT_MIN_FRAMES", "0")
    for (w, h, levels) in ((640, 480, 3), (328, 248, 3), (1920, 1080, 5)):
        a, b, _ = translated_pair(h, w, 41, max_shift=4)
        with H.FarnebackEngine(w, h, 1, levels=levels) as eng:
            np.testing.assert_array_equal(eng.calc(a, b), oracle.farneback(a, b, levels=levels, box_mode=oracle.BOX_BLOCKED))
    frames, _ = translated_pairs(20, 120, 160, 9100, max_shift=3)
    monkeypatch.setenv("OFARN_DIRECT_MIN_FRAMES", "1000000")
    with H.FarnebackEngine(160, 120, 20, levels=2) as eng:              # 40 frames in one wave through the row + column pass pair
        flow, _, _ = eng.calc_batch(frames, H.PAIRS_INDEPENDENT, want_danger=False)
    for i in (0, 7, 19):
        np.testing.assert_array_equal(flow[i], oracle.farneback(frames[2 * i], frames[2 * i + 1], levels=2, box_mode=oracle.BOX_BLOCKED))


# ------------------------------------------------------------------------------------ front end
@pytest.mark.parametrize("h,w", [(120, 160), (37, 53), (1, 1), (3, 5), (1080, 1920)])
def test_bgr2gray_bit_exact(H, oracle, h, w):
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    np.testing.assert_array_equal(H.cvtColor_bgr2gray(img), oracle.bgr2gray(img))
    with H.FarnebackEngine(w, h, 1) as eng:
        stack = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
        got = eng.bgr2gray(stack)
        for i in range(3):
            np.testing.assert_array_equal(got[i], oracle.bgr2gray(stack[i]))
        # a view with a row stride (cv2 accepts it): the mirror copies
        wide = rng.integers(0, 256, (h, w + 3, 3), dtype=np.uint8)
        np.testing.assert_array_equal(eng.bgr2gray(wide[:, :w]), oracle.bgr2gray(np.ascontiguousarray(wide[:, :w])))


def test_bgr_video_batch_device(H, oracle):
    torch = pytest.importorskip("torch")
    h, w, n_frames = 97, 131, 6         # odd sizes: frame offsets are not 4-byte aligned (byte kernel)
    rng = np.random.default_rng(77)
    gray, _ = translated_pairs(n_frames // 2, h, w, 5100, max_shift=3)
    # colour frames whose gray conversion is not trivially one channel
    bgr = np.stack([gray, np.roll(gray, 1, axis=2), 255 - gray], -1)
    bgr = (bgr.astype(np.int16) + rng.integers(-3, 4, bgr.shape)).clip(0, 255).astype(np.uint8)
    g_ref = np.stack([oracle.bgr2gray(f) for f in bgr])
    d_bgr = torch.from_numpy(bgr).cuda()
    for mode, n_pairs in ((H.PAIRS_CONSECUTIVE, n_frames - 1), (H.PAIRS_INDEPENDENT, n_frames // 2)):
        d_flow = torch.empty((n_pairs, h, w, 2), dtype=torch.float32, device="cuda")
        with H.FarnebackEngine(w, h, 2, levels=1) as eng:          # waves of 2 pairs: several waves, two streams
            eng.calc_batch_device(d_bgr, n_frames, w, h, mode, d_flow, None, None,
                                  stream=torch.cuda.current_stream().cuda_stream, bgr=True)
            torch.cuda.synchronize()
            d_gray = torch.empty((n_frames, h, w), dtype=torch.uint8, device="cuda")
            eng.bgr2gray_device(d_bgr, n_frames, w, h, d_gray, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        np.testing.assert_array_equal(d_gray.cpu().numpy(), g_ref)
        flow = d_flow.cpu().numpy()
        for i in range(n_pairs):
            a, b = (g_ref[i], g_ref[i + 1]) if mode == H.PAIRS_CONSECUTIVE else (g_ref[2 * i], g_ref[2 * i + 1])
            np.testing.assert_array_equal(flow[i], oracle.farneback(a, b, levels=1, box_mode=oracle.BOX_BLOCKED))


# ------------------------------------------------------------------------------------ USE_INITIAL_FLOW
@pytest.mark.parametrize("sw,sh,dw,dh", [(480, 270, 15, 9), (250, 130, 8, 5), (97, 83, 13, 11), (256, 128, 8, 4),
                                         (90, 60, 30, 20), (64, 32, 32, 16), (77, 33, 11, 33), (1920, 1080, 60, 34)])
def test_stage_resize_area_bit_exact(H, oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(sw + dh)
    src = (rng.standard_normal((sh, sw, 2)) * 3).astype(np.float32)
    with H.FarnebackEngine(sw, sh, 1) as eng:
        got = eng.stage_resize_area(src, dw, dh, 0.03125)
    np.testing.assert_array_equal(got, oracle.resize_area(src, dw, dh) * np.float32(0.03125))


@pytest.mark.parametrize("w,h,kw", [(320, 240, dict(levels=3)), (333, 251, dict(levels=2, winsize=9, iterations=2)),
                                    (160, 120, dict(levels=0)), (200, 150, dict(levels=1, iterations=1)),
                                    (256, 192, dict(levels=2, flags=256))])
def test_use_initial_flow(H, oracle, iter_kernel, w, h, kw, monkeypatch):
    kw = dict(kw)
    flags = kw.pop("flags", 0) | H.OPTFLOW_USE_INITIAL_FLOW
    a, b, (tx, ty) = translated_pair(h, w, 31, max_shift=4)
    rng = np.random.default_rng(6)
    init = (np.array([tx, ty], np.float32) + rng.standard_normal((h, w, 2)).astype(np.float32) * 0.5).astype(np.float32)
    ref = oracle.farneback(a, b, flags=flags, init_flow=init, box_mode=oracle.BOX_BLOCKED, **kw)
    buf = init.copy()
    got = H.calculate_optical_flow(a, b, buf, flags=flags, **kw)
    assert got is buf                                   # cv2 semantics: `flow` is written in place
    np.testing.assert_array_equal(got, ref)
    # the start matters (otherwise this test proves nothing)
    assert not np.array_equal(ref, oracle.farneback(a, b, flags=flags & 256, box_mode=oracle.BOX_BLOCKED, **kw))
    # unfused kernels take the same start
    monkeypatch.setenv("OFARN_FORCE_GENERIC", "1")
    with H.FarnebackEngine(w, h, 1, flags=flags, **kw) as eng:
        np.testing.assert_array_equal(eng.calc(a, b, init.copy()), ref)
    monkeypatch.delenv("OFARN_FORCE_GENERIC")
    with pytest.raises(ValueError):
        H.calculate_optical_flow(a, b, None, flags=flags, **kw)


def test_use_initial_flow_batch(H, oracle):
    torch = pytest.importorskip("torch")
    h, w, n_pairs = 120, 160, 5
    frames, shifts = translated_pairs(n_pairs, h, w, 6100, max_shift=4)
    init = np.zeros((n_pairs, h, w, 2), np.float32)
    init[...] = np.asarray(shifts, np.float32)[:, None, None, :]
    ref = [oracle.farneback(frames[2 * i], frames[2 * i + 1], levels=2, flags=4, init_flow=init[i], box_mode=oracle.BOX_BLOCKED)
           for i in range(n_pairs)]
    with H.FarnebackEngine(w, h, 2, levels=2, flags=4) as eng:
        flow, _, _ = eng.calc_batch(frames, H.PAIRS_INDEPENDENT, init_flow=init)
        for i in range(n_pairs):
            np.testing.assert_array_equal(flow[i], ref[i])
        d_flow = torch.from_numpy(init).cuda()                      # in/out on the device, several waves
        eng.calc_batch_device(torch.from_numpy(frames).cuda(), 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, None, None,
                              stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_flow.cpu().numpy(), np.stack(ref))
        with pytest.raises(ValueError):
            eng.calc_batch(frames, H.PAIRS_INDEPENDENT)
        with pytest.raises(ValueError):
            eng.calc_batch_device(torch.from_numpy(frames).cuda(), 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, None, None, None)


# ------------------------------------------------------------------------------------ visualisers
def test_hsv2bgr_bit_exact(H, oracle):
    Hh, S, V = np.meshgrid(np.arange(0, 256, 1), np.arange(0, 256, 5), np.arange(0, 256, 3), indexing="ij")
    hsv = np.stack([Hh, S, V], -1).astype(np.uint8).reshape(-1, 3)
    with H.FarnebackEngine(8, 8, 1) as eng:
        np.testing.assert_array_equal(eng.hsv2bgr(hsv), oracle.hsv2bgr_u8(hsv))


@pytest.mark.parametrize("h,w", [(120, 160), (1080, 1920)])
def test_draw_hsv(H, oracle, h, w):
    rng = np.random.default_rng(12)
    flow = (rng.standard_normal((h, w, 2)) * 20).astype(np.float32)
    flow[0, :8] = [(1, 0), (-1, 0), (0, 1), (0, -1), (0, 0), (-3, -0.0), (100, 100), (1e-20, -1e-20)]
    with H.FarnebackEngine(w, h, 1) as eng:
        bgr, hsv = eng.flow_hsv(flow, return_hsv=True)
    np.testing.assert_array_equal(H.draw_hsv(flow), bgr)
    # H, S, V and the BGR image all equal the NumPy lines with arctan2 correctly rounded + OpenCV's HSV2BGR arithmetic
    np.testing.assert_array_equal(hsv, oracle.draw_hsv_planes_numpy(flow))
    np.testing.assert_array_equal(bgr, oracle.draw_hsv_numpy(flow))


@pytest.mark.parametrize("h,w,step", [(120, 160, 14), (1080, 1920, 14), (100, 150, 15), (20, 20, 30), (5, 5, 14)])
def test_draw_flow_lines_bit_exact(H, oracle, h, w, step):
    rng = np.random.default_rng(13)
    flow = (rng.standard_normal((h, w, 2)) * 9).astype(np.float32)
    ref = oracle.draw_flow_lines_numpy((h, w), flow, step)
    got = H.flow_lines(flow, step)
    assert got.dtype == np.int32 and got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("h,w,step,radius", [(270, 480, 30, 6), (1080, 1920, 30, 6), (92, 100, 30, 6), (61, 75, 20, 9), (48, 64, 7, 3),
                                              (40, 90, 30, 0), (70, 130, 64, 31), (20, 20, 30, 6)])
def test_draw_sparse_lamps_bit_exact(H, oracle, h, w, step, radius):
    """The viewer's obstacle layer (pathfinder_viewer.py:196-222): every kept point of the danger map becomes a filled disc of colour
    (0, 0, V).  Reference side: the function's own NumPy lines on the kept int32 vectors and points, with cv2.cvtColor(HSV2BGR) and
    cv2.circle restated (parity unpinned, as everything OpenCV here); 92 x 100 and 70 x 130 clip discs at the image border."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(h * w + radius)
    flow = (rng.standard_normal((h, w, 2)) * rng.uniform(1, 40)).astype(np.float32)
    if len(oracle.grid_points_numpy(w, h, step)):
        mask, v, iflow = oracle.danger_map_numpy(flow, w, h, step, return_flow=True)
    else:                                                          # 20 x 20: no grid point fits, the layer is black
        mask, v, iflow = np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros((0, 2), np.int32)
    if len(mask) and not mask.any():
        mask[::2] = 1                                              # the filter kept nothing on this field: draw something anyway
        v[::2] = oracle.lamp_values_numpy(iflow[::2])
    pts = np.int32(oracle.grid_points_numpy(w, h, step) + 0.5)     # pathfinder_viewer.py:166
    keep = mask.astype(bool)
    ref = oracle.draw_sparse_lamps_numpy(iflow[keep], pts[keep], w, h, radius)
    got = H.draw_sparse_lamps(mask, v, (h, w), step=step, radius=radius)
    assert got.dtype == np.uint8 and got.shape == (h, w, 3)
    np.testing.assert_array_equal(got, ref)
    if keep.any():
        assert int(got[..., 2].max()) == int(v[keep].max()) and not got[..., :2].any()
    # cv2.add(output_bgr, layer) (pathfinder_viewer.py:299-300), as a stack of two frames
    base = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    with H.FarnebackEngine(w, h, 1, grid_step=step) as eng:
        both = eng.draw_lamps(np.stack([mask, mask[::-1]]), np.stack([v, v[::-1]]), (h, w), radius=radius, base=base)
        np.testing.assert_array_equal(both[0], oracle.cv_add_u8(base[0], ref))
        ref1 = oracle.draw_sparse_lamps_numpy(iflow[::-1][keep[::-1]], pts[keep[::-1]], w, h, radius)
        np.testing.assert_array_equal(both[1], oracle.cv_add_u8(base[1], ref1))
        # device-resident, unaligned output address (byte path)
        d_m, d_v = torch.from_numpy(mask).cuda(), torch.from_numpy(v).cuda()
        d_out = torch.zeros(h * w * 3 + 1, dtype=torch.uint8, device="cuda")
        eng.draw_lamps_device(d_m, d_v, 1, w, h, d_out[1:], radius=radius)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_out[1:].cpu().numpy().reshape(h, w, 3), ref)


@pytest.mark.parametrize("h,w,step,scale", [(120, 160, 14, 9), (1080, 1920, 14, 12), (100, 150, 15, 40), (33, 47, 7, 300), (20, 20, 30, 5),
                                             (64, 64, 1, 3)])
def test_draw_flow_image_bit_exact(H, oracle, h, w, step, scale):
    """draw_flow as the reference returns it (DenseOF.py:40-59): the BGR layer with every arrow rasterised as cv2.polylines draws it
    (clipLine + 8-connected LineIterator) and the radius-1 disc of cv2.circle at its start; flows long enough to leave the image
    exercise the clipping.  Reference side: the function's own lines with the cv2 calls restated (parity unpinned, like all of OpenCV)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(h + w + step)
    flow = (rng.standard_normal((h, w, 2)) * scale).astype(np.float32)
    flow[h // 2, w // 2] = (3e4, -2e4)
    ref = oracle.draw_flow_numpy((h, w), flow, step)
    got = H.draw_flow((h, w), flow, step)
    assert got.dtype == np.uint8 and got.shape == (h, w, 3)
    np.testing.assert_array_equal(got, ref)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    with H.FarnebackEngine(w, h, 1) as eng:
        np.testing.assert_array_equal(eng.draw_flow(flow, step, base=base), oracle.cv_add_u8(base, ref))   # DenseOF.py:574
        d_flow = torch.from_numpy(np.stack([flow, -flow])).cuda()
        d_img = torch.from_numpy(np.stack([base, base])).cuda()
        eng.draw_flow_device(d_flow, 2, w, h, d_img, step=step, d_base=d_img)                             # in place, two frames
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_img[0].cpu().numpy(), oracle.cv_add_u8(base, ref))
        np.testing.assert_array_equal(d_img[1].cpu().numpy(), oracle.cv_add_u8(base, oracle.draw_flow_numpy((h, w), -flow, step)))


@pytest.mark.parametrize("shape", [(1080, 1920, 3), (37, 53, 3), (5,), (0,), (64, 64)])
def test_add_u8_is_cv2_add(H, oracle, shape):
    """cv2.add on uint8 (how the viewers stack layers onto the frame, DenseOF.py:574-582): saturating, any size and alignment."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(int(np.prod(shape)) + 1)
    a, b = rng.integers(0, 256, shape, dtype=np.uint8), rng.integers(0, 256, shape, dtype=np.uint8)
    with H.FarnebackEngine(8, 8, 1) as eng:
        np.testing.assert_array_equal(eng.add_u8(a, b), oracle.cv_add_u8(a, b))
        if a.size > 3:
            d_a, d_b = torch.from_numpy(a.reshape(-1)).cuda(), torch.from_numpy(b.reshape(-1)).cuda()
            out = torch.zeros(a.size, dtype=torch.uint8, device="cuda")
            eng.add_u8_device(d_a[3:], d_b[3:], a.size - 3, out[3:])          # unaligned: byte path
            eng.add_u8_device(d_a, d_b, a.size, d_a)                         # aligned, in place
            torch.cuda.synchronize()
            ref = oracle.cv_add_u8(a, b).reshape(-1)
            np.testing.assert_array_equal(out[3:].cpu().numpy(), ref[3:])
            np.testing.assert_array_equal(d_a.cpu().numpy(), ref)


def test_draw_sparse_lamps_errors(H):
    with H.FarnebackEngine(64, 48, 1, grid_step=12) as eng:
        P = len(H.grid_points(64, 48, 12))
        z = np.zeros(P, np.uint8)
        with pytest.raises(NotImplementedError):
            eng.draw_lamps(z, z, (48, 64), radius=6)                # discs of radius 6 touch on a step-12 grid
        with pytest.raises(ValueError):
            eng.draw_lamps(z, z, (48, 64), radius=32)
        with pytest.raises(ValueError):
            eng.draw_lamps(z[:-1], z[:-1], (48, 64), radius=2)
        with pytest.raises(ValueError):
            eng.draw_lamps(z, z, (48, 64), radius=2, base=np.zeros((48, 64), np.uint8))
        assert not eng.draw_lamps(z, z, (48, 64), radius=5).any()


# ------------------------------------------------------------------------------------ sparse pyramidal LK
@pytest.mark.parametrize("h,w", [(37, 53), (270, 481), (1, 7), (6, 1), (1080, 1920)])
def test_lk_pyrdown_and_scharr_bit_exact(H, oracle, h, w):
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    with H.FarnebackEngine(w, h, 1) as eng:
        np.testing.assert_array_equal(eng.stage_pyrdown(img), oracle.pyrdown_u8(img))
        np.testing.assert_array_equal(eng.stage_scharr(img), oracle.scharr_deriv(img))


LK_CASES = [
    # (w, h, seed, points, kwargs)  -- first row: the call of pathfinder_viewer.py:156 / DenseOF.py:183
    (480, 270, 5, "grid30", dict(winSize=(45, 45), maxLevel=2, criteria=(3, 10, 0.03))),
    (320, 240, 6, "grid30", dict(winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))),          # SparseOF.py:6-8
    (333, 251, 7, "random", dict()),                                                              # cv2 defaults
    (200, 150, 8, "random", dict(winSize=(31, 17), maxLevel=1, criteria=(1, 5, 0.0))),             # COUNT only, non-square
    (200, 150, 9, "random", dict(winSize=(9, 9), maxLevel=4, criteria=(2, 0, 0.001))),             # EPS only -> 30 iterations
    (160, 120, 10, "random", dict(winSize=(21, 21), maxLevel=0, flags=8)),                         # OPTFLOW_LK_GET_MIN_EIGENVALS
    (100, 90, 11, "random", dict(winSize=(63, 63), maxLevel=3)),                                   # window barely fits level 0
]


def _lk_points(kind, w, h, oracle, seed):
    if kind == "grid30":
        return oracle.grid_points_numpy(w, h, 30)
    rng = np.random.default_rng(seed)
    pts = rng.uniform((-5, -5), (w + 5, h + 5), (300, 2)).astype(np.float32)
    pts[0] = (-200, 10)           # far outside: status 0
    pts[1] = (w - 1, h - 1)       # corner: window mostly in the border
    pts[2] = (0, 0)
    return pts


@pytest.mark.parametrize("w,h,seed,kind,kw", LK_CASES)
def test_lk_vs_oracle(H, oracle, w, h, seed, kind, kw):
    a, b, _ = translated_pair(h, w, seed, max_shift=5)
    pts = _lk_points(kind, w, h, oracle, seed)
    got_n, got_s, got_e = H.calcOpticalFlowPyrLK(a, b, pts, None, **kw)
    assert got_n.shape == pts.shape and got_s.shape == (len(pts), 1) and got_e.shape == (len(pts), 1)
    okw = dict(kw)
    ctype, cnt, eps = okw.pop("criteria", (3, 30, 0.01))
    okw["criteria"] = (cnt if ctype & 1 else 30, eps if ctype & 2 else 0.01)
    ref_n, ref_s, ref_e = oracle.calc_optical_flow_pyr_lk(a, b, pts, None, sum_mode=oracle.LK_SUM_COLUMNS, **okw)
    np.testing.assert_array_equal(got_s[:, 0], ref_s)
    np.testing.assert_array_equal(got_n, ref_n)
    np.testing.assert_array_equal(got_e[:, 0], ref_e)
    # OpenCV's scalar summation order: same status, positions within tolerance (float sums of ~2000 terms)
    sc_n, sc_s, _ = oracle.calc_optical_flow_pyr_lk(a, b, pts, None, sum_mode=oracle.LK_SUM_SCALAR, **okw)
    np.testing.assert_array_equal(got_s[:, 0], sc_s)
    d = np.abs(got_n - sc_n)[got_s[:, 0] == 1]
    if len(d):
        assert np.quantile(d, 0.99) < 5e-3 and d.max() < 0.1, (np.quantile(d, 0.99), d.max())


def test_lk_initial_flow_and_flat_image(H, oracle):
    a, b, (tx, ty) = translated_pair(150, 200, 21, max_shift=4)
    pts = np.random.default_rng(3).uniform((20, 20), (180, 130), (50, 2)).astype(np.float32)
    guess = (pts + (tx, ty) + 0.3).astype(np.float32)
    kw = dict(winSize=(15, 15), maxLevel=1, criteria=(3, 20, 0.01), flags=H.OPTFLOW_USE_INITIAL_FLOW)
    got = H.calcOpticalFlowPyrLK(a, b, pts, guess.copy(), **kw)
    ref = oracle.calc_optical_flow_pyr_lk(a, b, pts, guess.copy(), winSize=(15, 15), maxLevel=1, criteria=(20, 0.01),
                                          flags=oracle.LK_USE_INITIAL_FLOW, sum_mode=oracle.LK_SUM_COLUMNS)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1][:, 0], ref[1])
    with pytest.raises(ValueError):
        H.calcOpticalFlowPyrLK(a, b, pts, None, **kw)
    flat = np.full((150, 200), 90, np.uint8)
    n, s, e = H.calcOpticalFlowPyrLK(flat, flat, pts)
    assert not s.any() and np.array_equal(n, pts)
    with pytest.raises(ValueError):
        H.calcOpticalFlowPyrLK(a, b, pts, None, winSize=(2, 2))
    with pytest.raises(ValueError):
        H.calcOpticalFlowPyrLK(a, b[:, :-1], pts)


def test_lk_batch_device_and_get_flow_lk(H, oracle):
    torch = pytest.importorskip("torch")
    h, w, n_frames = 270, 480, 5
    frames, _ = translated_pairs(3, h, w, 7100, max_shift=5)
    frames = frames[:n_frames]
    pts = oracle.grid_points_numpy(w, h, 30)
    P = len(pts)
    d_frames = torch.from_numpy(frames).cuda()
    d_pts = torch.from_numpy(pts).cuda()
    lk = dict(winSize=(45, 45), maxLevel=2, criteria=(3, 10, 0.03))
    with H.FarnebackEngine(w, h, 2) as eng:                       # waves of 2 pairs
        for mode, n_pairs in ((H.PAIRS_CONSECUTIVE, n_frames - 1), (H.PAIRS_INDEPENDENT, n_frames // 2)):
            d_next = torch.zeros((n_pairs, P, 2), dtype=torch.float32, device="cuda")
            d_st = torch.zeros((n_pairs, P), dtype=torch.uint8, device="cuda")
            d_err = torch.zeros((n_pairs, P), dtype=torch.float32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream      # one stream for the library and for torch's own kernels
            eng.lk_batch_device(d_frames, n_frames if mode == H.PAIRS_CONSECUTIVE else 2 * n_pairs, w, h, mode, d_pts, P,
                                d_next, d_st, d_err, reverse=True, stream=st, **lk)
            d_mask = torch.zeros((n_pairs, P), dtype=torch.uint8, device="cuda")
            d_v = torch.zeros_like(d_mask)
            eng.vector_filter_device(d_next - d_pts, n_pairs, w, h, d_mask, d_v, stream=st)
            torch.cuda.synchronize()
            for i in range(n_pairs):
                f1, f2 = (frames[i], frames[i + 1]) if mode == H.PAIRS_CONSECUTIVE else (frames[2 * i], frames[2 * i + 1])
                # pathfinder_viewer.py:156: calcOpticalFlowPyrLK(img2, img1, ...) -- from the later frame to the earlier
                ref_n, ref_s, ref_e = oracle.calc_optical_flow_pyr_lk(f2, f1, pts, None, winSize=(45, 45), maxLevel=2,
                                                                      criteria=(10, 0.03), sum_mode=oracle.LK_SUM_COLUMNS)
                np.testing.assert_array_equal(d_next[i].cpu().numpy(), ref_n)
                np.testing.assert_array_equal(d_st[i].cpu().numpy(), ref_s)
                np.testing.assert_array_equal(d_err[i].cpu().numpy(), ref_e)
                mask, iflow, ipts, _ = oracle.get_flow_lk_numpy(f1, f2, pts, w, h, next_pts=ref_n)
                np.testing.assert_array_equal(d_mask[i].cpu().numpy().astype(bool), mask)
    # the reference's function, same name: (frame_layer, flow, points_)
    layer, flow, kept = H.get_flow_lk(frames[0], frames[1], pts)
    ref_n, _, _ = oracle.calc_optical_flow_pyr_lk(frames[1], frames[0], pts, None, winSize=(45, 45), maxLevel=2,
                                                  criteria=(10, 0.03), sum_mode=oracle.LK_SUM_COLUMNS)
    mask, iflow, ipts, _ = oracle.get_flow_lk_numpy(frames[0], frames[1], pts, w, h, next_pts=ref_n)
    assert flow.dtype == np.int32
    np.testing.assert_array_equal(kept, ipts[mask])
    np.testing.assert_array_equal(flow, iflow[mask])
    # the frame layer: kept vectors as red lines with magenta start circles; with the viewer's key 4 the rejected ones too
    np.testing.assert_array_equal(layer, oracle.get_flow_lk_layer_numpy(mask, iflow, ipts, w, h))
    layer_bad, _, _ = H.get_flow_lk(frames[0], frames[1], pts, draw_bad_flow=True)
    np.testing.assert_array_equal(layer_bad, oracle.get_flow_lk_layer_numpy(mask, iflow, ipts, w, h, draw_bad_flow=True))
    assert (layer_bad != layer).any() and set(map(tuple, layer.reshape(-1, 3))) <= {(0, 0, 0), (0, 0, 255), (255, 0, 255)}


@pytest.mark.parametrize("h,w,step,scale", [(270, 480, 30, 6), (1080, 1920, 30, 25), (61, 75, 20, 200), (40, 40, 3, 4)])
def test_draw_vectors_layer_bit_exact(H, oracle, h, w, step, scale):
    """get_flow_lk's drawing on arbitrary vectors: lines that cross, leave the image or have zero length, circles at the border; host
    and device entry points."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(h * 7 + w)
    pts = oracle.grid_points_numpy(w, h, step)
    vec = (rng.standard_normal((len(pts), 2)) * scale).astype(np.float32)
    vec[::7] = 0
    mask, _mod, iflow, ipts = oracle.vector_filter_numpy(vec, pts, w, h, 0)
    with H.FarnebackEngine(w, h, 1, grid_step=step) as eng:
        m2, _v, if2 = eng.vector_filter(vec, w, h, return_flow=True)
        np.testing.assert_array_equal(m2.astype(bool), mask)
        np.testing.assert_array_equal(if2, iflow)
        for bad in (False, True):
            ref = oracle.get_flow_lk_layer_numpy(mask, iflow, ipts, w, h, draw_bad_flow=bad)
            np.testing.assert_array_equal(eng.draw_vectors(if2, m2, (h, w), bad), ref)
        d_out = torch.empty((2, h, w, 3), dtype=torch.uint8, device="cuda")
        d_if = torch.from_numpy(np.stack([if2, -if2])).cuda()
        d_m = torch.from_numpy(np.stack([m2, 1 - m2])).cuda()
        eng.draw_vectors_device(d_if, d_m, 2, w, h, d_out, draw_bad_flow=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_out[0].cpu().numpy(), ref)
        np.testing.assert_array_equal(d_out[1].cpu().numpy(), oracle.get_flow_lk_layer_numpy(~mask, -iflow, ipts, w, h, draw_bad_flow=True))


def test_lk_1080p_grid(H, oracle):
    a, b, (tx, ty) = translated_pair(1080, 1920, 2001, max_shift=8)
    pts = H.grid_points(1920, 1080, 30)
    n, s, e = H.calcOpticalFlowPyrLK(b, a, pts, None, winSize=(45, 45), maxLevel=2, criteria=(3, 10, 0.03))
    ref_n, ref_s, ref_e = oracle.calc_optical_flow_pyr_lk(b, a, pts, None, winSize=(45, 45), maxLevel=2, criteria=(10, 0.03),
                                                          sum_mode=oracle.LK_SUM_COLUMNS)
    np.testing.assert_array_equal(n, ref_n)
    np.testing.assert_array_equal(s[:, 0], ref_s)
    inner = (pts[:, 0] > 100) & (pts[:, 0] < 1820) & (pts[:, 1] > 100) & (pts[:, 1] < 980)
    assert np.abs(n[inner] - pts[inner] + (tx, ty)).max() < 0.05          # tracked backwards: flow = -(tx, ty)


# ------------------------------------------------------------------------------------ the viewers' frame loop, headless
def test_headless_viewer_example(H, oracle):
    pytest.importorskip("torch")
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "headless_viewer", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "headless_viewer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    frames = mod.synthetic_video(5, 240, 320)
    out = mod.run(frames)
    P = len(out["points"])
    assert out["dense_mask"].shape == (4, P) and out["lk_mask"].shape == (4, P) and out["rainbow"].shape == (4, 240, 320, 3)
    # the scene pans by (2, 1) px per frame
    assert np.abs(out["mean_flow"] - (2, 1)).max() < 0.3          # whole-frame mean: the borders pull it down a little
    assert out["lk_status"].all()
    assert np.abs(np.median(out["lines"][:, :, 1] - out["lines"][:, :, 0], axis=1) - (-2, -1)).max() <= 1
    # the frame loop (one BGR frame per turn through FlowStream) gives what the device-resident batch gave
    flows, masks, vs = mod.run_loop(frames)
    np.testing.assert_array_equal(masks, out["dense_mask"])
    np.testing.assert_array_equal(vs, out["dense_v"])
    assert np.abs(flows.mean(axis=(1, 2)) - out["mean_flow"]).max() < 1e-4
    vm, vv, vl, shown = mod.run_loop_view(frames)
    np.testing.assert_array_equal(vm, out["dense_mask"])
    np.testing.assert_array_equal(vv, out["dense_v"])
    np.testing.assert_array_equal(vl, out["lines"])
    # the composited frame: the obstacle layer of each pair added onto the later frame (pathfinder_viewer.py:299-300)
    pts = np.int32(out["points"] + 0.5)
    for i in range(4):
        keep = out["dense_mask"][i].astype(bool)
        layer = np.zeros((240, 320, 3), np.uint8)
        for (x, y), val in zip(pts[keep], out["dense_v"][i][keep]):
            oracle.cv_circle_filled(layer, (x, y), 6, (0, 0, int(val)))
        np.testing.assert_array_equal(shown[i], oracle.cv_add_u8(frames[i + 1], layer))


def test_lk_batch_points_per_pair_and_forward_direction(H, oracle):
    torch = pytest.importorskip("torch")
    h, w, n_pairs, npts = 150, 200, 3, 40
    frames, _ = translated_pairs(n_pairs, h, w, 8100, max_shift=3)
    rng = np.random.default_rng(4)
    pts = rng.uniform((10, 10), (w - 10, h - 10), (n_pairs, npts, 2)).astype(np.float32)
    d_next = torch.zeros((n_pairs, npts, 2), dtype=torch.float32, device="cuda")
    d_st = torch.zeros((n_pairs, npts), dtype=torch.uint8, device="cuda")
    d_err = torch.zeros((n_pairs, npts), dtype=torch.float32, device="cuda")
    with H.FarnebackEngine(w, h, 2) as eng:
        eng.lk_batch_device(torch.from_numpy(frames).cuda(), 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, torch.from_numpy(pts).cuda(),
                            npts, d_next, d_st, d_err, reverse=False, pts_per_pair=True,
                            stream=torch.cuda.current_stream().cuda_stream, winSize=(21, 21), maxLevel=2)
        torch.cuda.synchronize()
    for i in range(n_pairs):
        ref_n, ref_s, ref_e = oracle.calc_optical_flow_pyr_lk(frames[2 * i], frames[2 * i + 1], pts[i], None, winSize=(21, 21),
                                                              maxLevel=2, sum_mode=oracle.LK_SUM_COLUMNS)
        np.testing.assert_array_equal(d_next[i].cpu().numpy(), ref_n)
        np.testing.assert_array_equal(d_st[i].cpu().numpy(), ref_s)
        np.testing.assert_array_equal(d_err[i].cpu().numpy(), ref_e)


# ------------------------------------------------------------------------------------ seeded fuzz over sizes and parameters
def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        w = int(rng.choice([rng.integers(33, 420), 64 * rng.integers(1, 6), 8 * rng.integers(5, 50)]))
        h = int(rng.choice([rng.integers(33, 300), 32 * rng.integers(2, 8), 8 * rng.integers(5, 36)]))
        kw = dict(levels=int(rng.integers(0, 5)), winsize=int(rng.choice([3, 5, 7, 8, 11, 15, 15, 15, 21, 25])),
                  iterations=int(rng.integers(1, 4)), poly_n=int(rng.choice([3, 5, 5, 5, 7])),
                  poly_sigma=float(rng.choice([1.1, 1.2, 1.5])), pyr_scale=float(rng.choice([0.5, 0.5, 0.5, 0.6, 0.8])),
                  flags=int(rng.choice([0, 0, 0, 4, 256, 260])))
        out.append((w, h, 9000 + i, kw))
    return out


@pytest.mark.parametrize("w,h,seed,kw", _fuzz_cases(36, 20261004))
def test_fuzz_pipeline_bit_exact(H, oracle, iter_kernel, w, h, seed, kw, monkeypatch):
    """Random sizes and parameter sets (fused, generic, direct-level and flag paths all get hit): the whole pipeline
    bit for bit against the oracle in the device summation order."""
    monkeypatch.setenv("OFARN_DIRECT_MIN_FRAMES", "0" if seed % 2 else "1000000")
    if seed % 3 == 0:      # a third of the cases on the FPV-like warped family (non-uniform, sub-pixel flow + occluder)
        a, b, gt, _ = warped_pair(h, w, seed, zoom=1.0 + 0.01 * (seed % 5), angle_deg=(seed % 7) - 3.0)
        tx, ty = (float(v) for v in gt[h // 2, w // 2])
    else:
        a, b, (tx, ty) = translated_pair(h, w, seed, max_shift=4)
    init = None
    if kw["flags"] & 4:
        init = np.empty((h, w, 2), np.float32)
        init[...] = (tx + 0.25, ty - 0.5)
    ref = oracle.farneback(a, b, box_mode=oracle.BOX_BLOCKED, init_flow=init, **kw)
    with H.FarnebackEngine(w, h, 1, **kw) as eng:
        got = eng.calc(a, b, None if init is None else init.copy())
    np.testing.assert_array_equal(got, ref)


def _lk_fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        w, h = int(rng.integers(40, 400)), int(rng.integers(40, 300))
        win = (int(rng.integers(3, 50)), int(rng.integers(3, 50)))
        kw = dict(winSize=win, maxLevel=int(rng.integers(0, 5)),
                  criteria=(int(rng.choice([1, 2, 3])), int(rng.integers(1, 25)), float(rng.choice([0.001, 0.01, 0.03, 0.3]))),
                  flags=int(rng.choice([0, 0, 0, 8])), minEigThreshold=float(rng.choice([1e-4, 1e-3, 1e-6])))
        out.append((w, h, 9500 + i, kw))
    return out


@pytest.mark.parametrize("w,h,seed,kw", _lk_fuzz_cases(24, 7))
def test_fuzz_lk_bit_exact(H, oracle, w, h, seed, kw):
    if seed % 2:
        a, b, _, _ = warped_pair(h, w, seed, zoom=1.02, angle_deg=1.0)
    else:
        a, b, _ = translated_pair(h, w, seed, max_shift=4)
    pts = np.random.default_rng(seed).uniform((-8, -8), (w + 8, h + 8), (120, 2)).astype(np.float32)
    got_n, got_s, got_e = H.calcOpticalFlowPyrLK(a, b, pts, None, **kw)
    okw = dict(kw)
    ctype, cnt, eps = okw.pop("criteria")
    okw["criteria"] = (cnt if ctype & 1 else 30, eps if ctype & 2 else 0.01)
    ref_n, ref_s, ref_e = oracle.calc_optical_flow_pyr_lk(a, b, pts, None, sum_mode=oracle.LK_SUM_COLUMNS, **okw)
    np.testing.assert_array_equal(got_s[:, 0], ref_s)
    np.testing.assert_array_equal(got_n, ref_n)
    np.testing.assert_array_equal(got_e[:, 0], ref_e)


def test_torch_default_stream_is_respected(H, oracle):
    """torch's default stream has handle 0; the Python mirror must enqueue on THAT stream (not on the engine's own), so work
    torch has queued but not finished -- here the copy that fills the frames behind a long matmul -- is seen by the kernels."""
    torch = pytest.importorskip("torch")
    h, w = 120, 160
    a, b, _ = translated_pair(h, w, 91, max_shift=3)
    src = torch.from_numpy(np.stack([a, b])).cuda()
    d_frames = torch.zeros_like(src)
    d_flow = torch.zeros((1, h, w, 2), dtype=torch.float32, device="cuda")
    big = torch.randn((6144, 6144), device="cuda")
    with H.FarnebackEngine(w, h, 1, levels=2) as eng:
        eng.calc_batch_device(src, 2, w, h, H.PAIRS_INDEPENDENT, d_flow, None, None)   # warm-up: builds the plan (which synchronises)
        torch.cuda.synchronize()
        d_flow.zero_()
        torch.cuda.synchronize()
        for _ in range(6):
            big = big @ big * 1e-3                      # keeps the default stream busy for a while
        d_frames.copy_(src)                             # queued behind the matmuls
        eng.calc_batch_device(d_frames, 2, w, h, H.PAIRS_INDEPENDENT, d_flow, None, None,
                              stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    np.testing.assert_array_equal(d_flow[0].cpu().numpy(), oracle.farneback(a, b, levels=2, box_mode=oracle.BOX_BLOCKED))


# ------------------------------------------------------------------------------------ FPV-like warped family
WARP_CASES = [
    # (w, h, seed, warp kwargs, farneback kwargs)
    (640, 480, 31, dict(), dict()),                                            # DenseOF.py defaults
    (480, 270, 32, dict(zoom=1.05, angle_deg=-2.0), dict(levels=3)),
    (333, 251, 33, dict(zoom=0.97, angle_deg=3.0, shift=(-2.3, 1.1)), dict(levels=2, winsize=9, iterations=2)),   # receding
    (320, 240, 34, dict(zoom=1.08, occluder=False), dict(levels=3, winsize=21)),
    (300, 200, 35, dict(focus=(40, 30)), dict(levels=2, poly_n=7, poly_sigma=1.5)),   # focus of expansion off-centre
    (256, 192, 36, dict(zoom=1.02), dict(levels=2, flags=256)),                      # Gaussian window
    (1920, 1080, 37, dict(zoom=1.02, angle_deg=0.5), dict(levels=5)),                  # BASELINE config 2 shape
]


@pytest.mark.parametrize("w,h,seed,wkw,kw", WARP_CASES)
def test_pipeline_warped_family(H, oracle, iter_kernel, w, h, seed, wkw, kw):
    """Zoom + rotation + sub-pixel shift + occluding patch: the flow-dependent gather of FarnebackUpdateMatrices is
    unaligned and, along the borders the flow points out of, takes its out-of-image branch row after row."""
    a, b, gt, valid = warped_pair(h, w, seed, **wkw)
    got = H.calculate_optical_flow(a, b, **kw)
    ref = oracle.farneback(a, b, box_mode=oracle.BOX_BLOCKED, **kw)
    np.testing.assert_array_equal(got, ref)
    if not kw.get("flags", 0):
        e = epe(got, oracle.farneback(a, b, box_mode=oracle.BOX_RUNNING, **kw))
        if kw.get("winsize", 15) >= 15:
            assert e.mean() <= TOL_MEAN_EPE and e.max() <= TOL_MAX_EPE, (e.mean(), e.max())
        else:
            assert e.mean() <= 1e-4 and np.quantile(e, 0.999) <= 1e-3 and e.max() <= 0.5, (e.mean(), e.max())
    # the out-of-image branch really fires: some pixels' flow points outside the frame
    ys, xs = np.mgrid[0:h, 0:w]
    out = (xs + got[..., 0] < 0) | (xs + got[..., 0] >= w - 1) | (ys + got[..., 1] < 0) | (ys + got[..., 1] >= h - 1)
    if wkw.get("zoom", 1.03) > 1.0:
        assert out.sum() > 0
    # and the estimate is a flow estimate: close to the ground truth away from the borders and the patch
    inner = valid.copy()
    m = max(16, min(h, w) // 8)
    inner[:m] = inner[-m:] = False
    inner[:, :m] = inner[:, -m:] = False
    assert epe(got, gt)[inner].mean() < 0.5


def test_warped_batch_danger_sets(H, oracle):
    """Batch of warped pairs through the device entry point: flow bit-exact, danger maps equal to the reference's NumPy
    filter on the same flow, and the danger set follows the radial field (points far from the focus move most;
    the equalisation of pathfinder_viewer.py:162-168 is what keeps them from all being selected)."""
    h, w, n = 270, 480, 4
    frames, flows = warped_pairs(n, h, w, 7300)
    with H.FarnebackEngine(w, h, 3, levels=3) as eng:
        flow, mask, v = eng.calc_batch(frames, H.PAIRS_INDEPENDENT)
    for i in range(n):
        np.testing.assert_array_equal(flow[i], oracle.farneback(frames[2 * i], frames[2 * i + 1], levels=3, box_mode=oracle.BOX_BLOCKED))
        m_ref, v_ref = oracle.danger_map_numpy(flow[i], w, h, 30)
        np.testing.assert_array_equal(mask[i], m_ref)
        np.testing.assert_array_equal(v[i], v_ref)
        assert 0 < mask[i].sum() < mask[i].size // 2 + 1


def test_danger_sets_gpu_flow_vs_opencv_order_flow(H, oracle):
    """north_star asks for bit-identical danger-point index sets.  On the SAME flow the mask is bit-exact (tests above).
    This measures the other comparison: the mask from the GPU's flow against the mask from the oracle's flow in OpenCV's
    literal running-sum order (the two flows differ by ~1e-6 px).  A point flips only if its equalised modulus sits
    within that distance of the median or the 99-percentile; the count is reported and must be 0 on these 12 pairs."""
    h, w = 270, 480
    sym = 0
    total = 0
    pairs = [translated_pair(h, w, 8800 + i, max_shift=6)[:2] for i in range(6)] + \
            [warped_pair(h, w, 8900 + i, zoom=1.01 + 0.01 * i)[:2] for i in range(6)]
    for a, b in pairs:
        got = H.calculate_optical_flow(a, b, levels=3)
        lit = oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_RUNNING)
        m_gpu, _ = H.danger_map(got, 30)
        m_lit, _ = oracle.danger_map_numpy(lit, w, h, 30)
        sym += int((m_gpu != m_lit).sum())
        total += m_gpu.size
    print(f"danger-set symmetric difference, GPU flow vs OpenCV-order flow: {sym} of {total} grid points")
    assert sym == 0


def test_trig_outputs_vs_literal_numpy(H, oracle):
    """V, the integer vectors and the hue on >= 10^6 random vectors: equal to the NumPy lines with correctly rounded
    arctan2 / cos / sin (the contract), and within a measured rate of the LITERAL lines on this machine's NumPy, whose
    float32 arctan2 is a SIMD approximation (<= 3.2 ulp here, 0.5 / 3.5 mismatches per 10^6 after truncation measured in
    the build container, 0 and 1 on the GPU box; the bound is 5 per 10^6 for both)."""
    rng = np.random.default_rng(42)
    w, h, step = 1920, 1080, 5                      # 82 944 grid points per map (also: P far beyond one LDS sort)
    pts = oracle.grid_points_numpy(w, h, step)
    P = len(pts)
    n_maps = 13                                     # 1.08 x 10^6 vectors
    vec = (rng.standard_normal((n_maps, P, 2)) * 5).astype(np.float32)
    bad_cr = bad_np = 0
    with H.FarnebackEngine(w, h, 1, grid_step=step) as eng:
        mask, v, iflow = eng.vector_filter(vec, w, h, return_flow=True)
    for i in range(n_maps):
        m_ref, _, if_cr, _ = oracle.vector_filter_numpy(vec[i], pts, w, h, cr=True)
        _, _, if_np, _ = oracle.vector_filter_numpy(vec[i], pts, w, h, cr=False)
        np.testing.assert_array_equal(mask[i].astype(bool), m_ref)
        bad_cr += int((iflow[i] != if_cr).any(axis=1).sum())
        bad_np += int((iflow[i] != if_np).any(axis=1).sum())
    print(f"integer vectors: {bad_cr} differ from correctly rounded NumPy, {bad_np} from literal NumPy, of {n_maps * P}")
    assert bad_cr == 0
    assert bad_np <= 5 * n_maps * P // 1_000_000
    flow = vec[:12].reshape(864, 1152, 2)           # 10^6 pixels through draw_hsv
    with H.FarnebackEngine(1152, 864, 1) as eng:
        _, hsv = eng.flow_hsv(flow, return_hsv=True)
    np.testing.assert_array_equal(hsv, oracle.draw_hsv_planes_numpy(flow, cr=True))
    bad_h = int((hsv[..., 0] != oracle.draw_hsv_planes_numpy(flow, cr=False)[..., 0]).sum())
    print(f"hue: {bad_h} of {flow.shape[0] * flow.shape[1]} differ from literal NumPy")
    assert bad_h <= 5 * flow.shape[0] * flow.shape[1] // 1_000_000


def test_fine_grid_and_nan(H, oracle):
    """a fine grid (step 5 at 1080p = 82 944 points) used to need more LDS than a block has and returned
    garbage silently; the radix select has no such limit.  Also: a NaN vector empties the mask, as NumPy's NaN median does."""
    rng = np.random.default_rng(5)
    w, h = 1920, 1080
    flow = (rng.standard_normal((h, w, 2)) * 3).astype(np.float32)
    for step in (5, 7, 30, 200, 1000):
        mask, v = H.danger_map(flow, step)
        m_ref, v_ref = oracle.danger_map_numpy(flow, w, h, step)
        assert mask.shape == m_ref.shape
        np.testing.assert_array_equal(mask, m_ref)
        np.testing.assert_array_equal(v, v_ref)
    flow[15, 15] = np.nan
    mask, v = H.danger_map(flow, 30)
    assert not mask.any() and not v.any()


def test_calls_on_different_streams_are_ordered(H, oracle):
    """every entry point of a context shares one workspace.  A device call on torch's stream followed at once
    by a host call (the context's own stream) must not overwrite R / flow buffers the first one is still using."""
    torch = pytest.importorskip("torch")
    h, w, n_pairs = 270, 480, 24
    frames, _ = translated_pairs(n_pairs, h, w, 4400, max_shift=4, unique=4)
    a, b, _ = translated_pair(h, w, 4499, max_shift=5)
    d_frames = torch.from_numpy(frames).cuda()
    d_flow = torch.zeros((n_pairs, h, w, 2), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    refs = [oracle.farneback(frames[2 * i], frames[2 * i + 1], levels=3, box_mode=oracle.BOX_BLOCKED) for i in range(4)]
    ref_single = oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_BLOCKED)
    with H.FarnebackEngine(w, h, n_pairs, levels=3) as eng:
        eng.calc(a, b)                                   # builds the plan
        torch.cuda.synchronize()
        for _ in range(3):
            eng.calc_batch_device(d_frames, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, None, None, stream=side.cuda_stream)
            single = eng.calc(a, b)                      # own stream, issued while the batch is still running
            side.synchronize()
            np.testing.assert_array_equal(single, ref_single)
            got = d_flow.cpu().numpy()
            for i in range(n_pairs):
                np.testing.assert_array_equal(got[i], refs[i % 4])


def test_real_opencv_if_present(H):
    """The guarded real-OpenCV column: cv2 is not installed in the build container nor, as far as is
    known, on the GPU box; if it ever is, this is the one test that pins the pipeline to the reference's real
    dependency.  Nothing else depends on it."""
    cv2 = pytest.importorskip("cv2")
    for a, b in (translated_pair(480, 640, 1001)[:2], warped_pair(480, 640, 1002)[:2]):
        ref = cv2.calcOpticalFlowFarneback(a, b, None, 0.5, 3, 15, 3, 5, 1.2, 0)
        got = H.calculate_optical_flow(a, b)
        e = epe(got, ref)
        print(f"EPE vs real OpenCV {cv2.__version__}: mean {e.mean():.3e} max {e.max():.3e}")
        assert e.mean() <= 1e-4 and np.quantile(e, 0.999) <= 1e-2


def test_bench_two_gloo_ranks_share_the_gpu():
    """Rehearsal of the multi-rank path on the one-GPU box: bench.py starts its own two ranks (fresh children through
    torch.distributed.run), both use cuda:0, pairs are sharded 512 -> here 16 in total (strong scaling, config 4 shape),
    and the danger maps are all-gathered (through host memory under gloo).  The log is kept under gpurun_out/."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "4", "--batch", "16",
           "--steps", "2", "--warmup", "1", "--cpu-sample", "0"]
    env = dict(os.environ, OFARN_BENCH_SHARE_GPU="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "bench_gloo2.log"), "w") as f:
        f.write("$ " + " ".join(cmd) + "\n" + r.stdout + "\n---- stderr ----\n" + r.stderr[-4000:])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # two ranks on ONE device: the line reports the distinct devices as n_gpus and the ranks separately
    assert out["n_gpus"] == 1 and out["ranks"] == 2 and out["scaling"] == "strong" and out["config"]["global_pairs"] == 16
    assert "share a GPU" in out["config"]["parallelism"]
    assert out["config"]["pairs_per_gpu"] == 8 and out["value"] > 0
    assert out["gathered_danger_maps_checked"] is True


@pytest.mark.parametrize("config,batch", [(4, 16), (5, 2)])
def test_bench_rccl_leg_at_world_size_one(config, batch):
    """The RCCL leg of BASELINE configs 4 and 5 as far as ONE GPU allows: `bench.py --gpus 1 --backend nccl --force-dist` starts
    its rank as a fresh child through torch.distributed.run, initialises the `nccl` (= RCCL) process group at world size 1 and
    runs the allocation-free all_gather_into_tensor of the danger maps on DEVICE tensors inside the timed region; afterwards the
    rank checks its shard at its place in the gathered arrays.  (Two nccl ranks on one GPU are not possible; the 2-rank
    layout is covered by the gloo rehearsal above and by tests/test_distributed_cpu.py.)  The log is kept under gpurun_out/."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-dist", "--config", str(config),
           "--batch", str(batch), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-family-check"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", f"bench_rccl_ws1_config{config}.log"), "w") as f:
        f.write("$ " + " ".join(cmd) + "\n" + r.stdout + "\n---- stderr ----\n" + r.stderr[-4000:])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["ranks"] == 1 and out["scaling"] == "strong" and out["config"]["global_pairs"] == batch
    assert out["gathered_danger_maps_checked"] is True
    assert out["collective"]["backend"] == "nccl" and out["collective"]["device"].startswith("cuda")
    assert out["collective"]["calls"] >= 3 and out["value"] > 0
    assert out["roofline"]["frac"] <= 1.0


def test_workspace_grows_on_demand_and_fails_cleanly(H, oracle, monkeypatch):
    """R and the flow buffers are reserved by ofarn_create, the level-image / row-pass / matrix buffers by the first call that
    needs them (round 1 reserved all of them at full resolution: 207 MB per 1080p pair, now 123).  A context that cannot get
    its memory fails with MemoryError (OFARN_E_NOMEM), not with a crash."""
    a, b, _ = translated_pair(240, 320, 61, max_shift=3)
    c, d, _ = translated_pair(480, 640, 62, max_shift=3)
    with H.FarnebackEngine(640, 480, 2, levels=3) as eng:
        w0 = eng.workspace_bytes
        assert w0 >= 2 * 2 * (640 * 480 * 5 + 4) * 4 + 2 * 2 * 640 * 480 * 2 * 4      # R for 4 frames + two flow buffers for 2 pairs
        assert w0 < 2 * 207e6 * (640 * 480) / (1920 * 1080)                           # nothing else yet
        np.testing.assert_array_equal(eng.calc(a, b), oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_BLOCKED))
        w1 = eng.workspace_bytes
        np.testing.assert_array_equal(eng.calc(c, d), oracle.farneback(c, d, levels=3, box_mode=oracle.BOX_BLOCKED))
        w2 = eng.workspace_bytes
        np.testing.assert_array_equal(eng.calc(a, b), oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_BLOCKED))
        assert w0 < w1 < w2 == eng.workspace_bytes                                    # grows with the shapes seen, never shrinks
        M = np.random.default_rng(0).standard_normal((480, 640, 5)).astype(np.float32)
        eng.stage_blur_solve(M)                                                       # the unfused stage needs M: allocated now
        assert eng.workspace_bytes > w2
    # the unfused path (OFARN_FORCE_GENERIC) allocates M for the batch and still matches
    monkeypatch.setenv("OFARN_FORCE_GENERIC", "1")
    with H.FarnebackEngine(640, 480, 1, levels=3) as eng:
        np.testing.assert_array_equal(eng.calc(c, d), oracle.farneback(c, d, levels=3, box_mode=oracle.BOX_BLOCKED))
    monkeypatch.delenv("OFARN_FORCE_GENERIC")
    with pytest.raises(MemoryError):
        H.FarnebackEngine(3840, 2160, 4000)          # 4000 pairs of 4K: ~2 TB of R alone


def test_c_program_through_the_c_abi(H, tmp_path):
    """The boundary without Python: examples/c_abi_pair.c (plain C99, links libofarn.so) reads two raw frames, calls ofarn_create /
    ofarn_calc / ofarn_grid_filter and writes the flow; it must equal the Python mirror's result bit for bit."""
    import subprocess
    from test_host_abi import _build_c_example
    exe = _build_c_example(tmp_path)
    h, w = 270, 480
    a, b, _, _ = warped_pair(h, w, 4321)
    (tmp_path / "frames.raw").write_bytes(a.tobytes() + b.tobytes())
    r = subprocess.run([exe, str(tmp_path / "frames.raw"), str(w), str(h), str(tmp_path / "flow.raw"), "3"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    flow_c = np.fromfile(tmp_path / "flow.raw", np.float32).reshape(h, w, 2)
    flow_py = H.calculate_optical_flow(a, b, levels=3)
    np.testing.assert_array_equal(flow_c, flow_py)
    mask, _ = H.danger_map(flow_py, 30)
    assert f"danger points: {int(mask.sum())} of {mask.size} grid points" in r.stdout


def test_contexts_give_their_memory_back(H):
    """ofarn_destroy frees everything a context ever allocated: workspace (eager and grown on demand), staging, plan tables,
    the LK pyramid, streams and events.  Free device memory after 20 create / use / destroy cycles equals what it was before."""
    torch = pytest.importorskip("torch")
    a, b, _ = translated_pair(240, 320, 71, max_shift=3)
    frames, _ = translated_pairs(3, 240, 320, 72, max_shift=3)
    pts = H.grid_points(320, 240, 30)
    H.close_cached_engines()

    def cycle(flags):
        with H.FarnebackEngine(320, 240, 2, levels=2, flags=flags) as eng:
            eng.calc(a, b, np.zeros((240, 320, 2), np.float32) if flags & 4 else None)
            eng.calc_batch(frames, H.PAIRS_INDEPENDENT, init_flow=np.zeros((3, 240, 320, 2), np.float32) if flags & 4 else None)
            eng.danger_map(np.zeros((240, 320, 2), np.float32))
            eng.lk(a, b, pts)
            eng.bgr2gray(np.zeros((240, 320, 3), np.uint8))
            eng.stage_blur_solve(np.zeros((240, 320, 5), np.float32))

    cycle(0)                                  # first use: module load, runtime pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(20):
        cycle((0, 4, 256, 260)[i % 4])
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) <= 8 << 20, (free0, free1)       # nothing held back (allocator granularity aside)


def test_drop_in_call_from_several_threads(H, oracle):
    """the cache behind calculate_optical_flow() holds 8 contexts and used to close the oldest while another thread
    could be inside it.  Four threads call the drop-in function with twelve different frame sizes (so entries are evicted all the
    time) and with shared sizes (so threads meet on one context): every result must be right and nothing may crash."""
    import threading
    sizes = [(96 + 8 * i, 64 + 8 * (i % 5)) for i in range(12)]
    pairs = {s: translated_pair(s[1], s[0], 500 + i, max_shift=3)[:2] for i, s in enumerate(sizes)}
    refs = {s: oracle.farneback(p[0], p[1], levels=1, box_mode=oracle.BOX_BLOCKED) for s, p in pairs.items()}
    errors = []

    def worker(k):
        try:
            rng = np.random.default_rng(k)
            for _ in range(30):
                s = sizes[int(rng.integers(0, len(sizes)))]
                got = H.calculate_optical_flow(pairs[s][0], pairs[s][1], levels=1)
                if not np.array_equal(got, refs[s]):
                    errors.append(("mismatch", s))
        except Exception as e:      # noqa: BLE001
            errors.append(("exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    H.close_cached_engines()


@pytest.mark.parametrize("w,h,kw", [
    (4096, 96, dict(levels=1)),                       # wide and flat
    (96, 4096, dict(levels=1)),                       # narrow and tall: many strips, one column block
    (300, 200, dict(levels=4, pyr_scale=0.3)),        # steep pyramid: cropped to the levels that stay >= 32 px
    (300, 200, dict(levels=3, pyr_scale=0.95)),       # nearly flat pyramid: no level is a power-of-two fraction
    (200, 150, dict(levels=2, iterations=0)),         # no iterations: the flow is only initialised and upsampled
    (200, 150, dict(levels=0, iterations=6)),         # one scale, many iterations
    (257, 131, dict(levels=2, winsize=2)),            # smallest window
    (160, 120, dict(levels=1, poly_n=1, poly_sigma=0.0)),   # smallest polynomial support, sigma from n
])
def test_unusual_shapes_and_parameters(H, oracle, w, h, kw):
    a, b, _ = translated_pair(h, w, 600 + w % 97, max_shift=3)
    got = H.calculate_optical_flow(a, b, **kw)
    np.testing.assert_array_equal(got, oracle.farneback(a, b, box_mode=oracle.BOX_BLOCKED, **kw))


def test_two_contexts_from_two_threads(H, oracle):
    """Contexts are independent: two threads, each with its own engine (different parameters, same GPU), run at the same time."""
    import threading
    a, b, _ = translated_pair(240, 320, 81, max_shift=3)
    frames, _ = translated_pairs(6, 240, 320, 82, max_shift=3)
    ref1 = oracle.farneback(a, b, levels=2, box_mode=oracle.BOX_BLOCKED)
    ref2 = [oracle.farneback(frames[2 * i], frames[2 * i + 1], levels=3, winsize=9, box_mode=oracle.BOX_BLOCKED) for i in range(6)]
    errors = []

    def single():
        try:
            with H.FarnebackEngine(320, 240, 1, levels=2) as eng:
                for _ in range(15):
                    if not np.array_equal(eng.calc(a, b), ref1):
                        errors.append("single")
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    def batch():
        try:
            with H.FarnebackEngine(320, 240, 4, levels=3, winsize=9) as eng:
                for _ in range(5):
                    flow, _, _ = eng.calc_batch(frames, H.PAIRS_INDEPENDENT, want_danger=False)
                    if not all(np.array_equal(flow[i], ref2[i]) for i in range(6)):
                        errors.append("batch")
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=single), threading.Thread(target=batch)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_device_batch_is_graph_capturable(H, oracle):
    """After one warm-up call (which builds the plan and grows the workspace) a device-resident batch call allocates nothing
    and synchronises nothing, so it can be captured into a HIP graph and replayed -- including the fork / join over the two
    internal streams that a batch of several waves uses.  The replayed graph must give the same flow and danger maps."""
    torch = pytest.importorskip("torch")
    h, w, n_pairs = 120, 160, 6
    frames, _ = translated_pairs(n_pairs, h, w, 4700, max_shift=4)
    other, _ = translated_pairs(n_pairs, h, w, 4800, max_shift=4)
    P = len(H.grid_points(w, h, 30))
    d_frames = torch.from_numpy(frames).cuda()
    d_flow = torch.zeros((n_pairs, h, w, 2), dtype=torch.float32, device="cuda")
    d_mask = torch.zeros((n_pairs, P), dtype=torch.uint8, device="cuda")
    d_v = torch.zeros_like(d_mask)
    side = torch.cuda.Stream()
    with H.FarnebackEngine(w, h, 4, levels=2) as eng:          # waves of 4 pairs: two waves, two internal streams
        with torch.cuda.stream(side):
            eng.calc_batch_device(d_frames, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, d_mask, d_v, stream=side.cuda_stream)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            eng.calc_batch_device(d_frames, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, d_mask, d_v, stream=side.cuda_stream)
        for src in (other, frames):
            d_frames.copy_(torch.from_numpy(src).cuda())
            d_flow.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            got = d_flow.cpu().numpy()
            for i in range(n_pairs):
                np.testing.assert_array_equal(got[i], oracle.farneback(src[2 * i], src[2 * i + 1], levels=2, box_mode=oracle.BOX_BLOCKED))
                m_ref, v_ref = oracle.danger_map_numpy(got[i], w, h, 30)
                np.testing.assert_array_equal(d_mask[i].cpu().numpy(), m_ref)
                np.testing.assert_array_equal(d_v[i].cpu().numpy(), v_ref)
        del g


def test_reserve_then_capture_and_capture_guard(H, oracle):
    """a call recorded into a HIP graph must not allocate.  A fresh context whose workspace would have to grow
    during capture refuses (ValueError) instead of freeing / allocating buffers that a graph would keep replaying into;
    ofarn_reserve grows everything up front (level plan, row-pass / level-image buffers, second workspace), after which the very
    first call can be captured and replays bit-exactly."""
    torch = pytest.importorskip("torch")
    w, h, n_pairs = 200, 150, 6
    frames, _ = translated_pairs(n_pairs, h, w, 4100, max_shift=3)
    d_frames = torch.from_numpy(frames).cuda()
    d_flow = torch.empty((n_pairs, h, w, 2), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with H.FarnebackEngine(w, h, 4, levels=2) as eng:
        g = torch.cuda.CUDAGraph()
        with pytest.raises(ValueError, match="captured"):
            with torch.cuda.graph(g, stream=side):
                eng.calc_batch_device(d_frames, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, stream=side.cuda_stream)
        del g
        torch.cuda.synchronize()
    with H.FarnebackEngine(w, h, 4, levels=2) as eng:
        w0 = eng.workspace_bytes
        eng.reserve(w, h, n_pairs)
        w1 = eng.workspace_bytes
        assert w1 > w0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            eng.calc_batch_device(d_frames, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, stream=side.cuda_stream)
        assert eng.workspace_bytes == w1                     # nothing grew during capture
        d_flow.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        got = d_flow.cpu().numpy()
        for i in range(n_pairs):
            np.testing.assert_array_equal(got[i], oracle.farneback(frames[2 * i], frames[2 * i + 1], levels=2, box_mode=oracle.BOX_BLOCKED))
        del g
        with pytest.raises(ValueError):
            eng.reserve(w, h, 0)
        with pytest.raises(ValueError):
            eng.set_option("no_such_option", 1)


@pytest.mark.parametrize("fail_wave", [0, 1, 2])
def test_error_in_the_middle_of_a_batch_leaves_the_context_usable(H, oracle, fail_wave):
    """an error in wave k of a multi-wave batch (the waves alternate over two internal streams) used to
    return without joining those streams or recording the call's event, so the next call could race the still-running waves
    on the shared workspace.  The failing wave is injected (ofarn_set_option "debug_fail_wave": the kernels of the earlier
    waves are already enqueued when it fires); the call reports MemoryError, and the NEXT call on the same context -- on
    another stream, right away -- is correct for every pair."""
    torch = pytest.importorskip("torch")
    w, h, n_pairs = 320, 240, 6
    frames, _ = translated_pairs(n_pairs, h, w, 4200, max_shift=3)
    other, _ = warped_pairs(n_pairs, h, w, 4300)
    d_a = torch.from_numpy(frames).cuda()
    d_b = torch.from_numpy(other).cuda()
    d_flow = torch.empty((n_pairs, h, w, 2), dtype=torch.float32, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with H.FarnebackEngine(w, h, 2, levels=3) as eng:          # waves of 2 pairs: 3 waves over the two internal streams
        eng.calc_batch_device(d_a, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, stream=s1.cuda_stream)   # warm: workspaces exist
        s1.synchronize()
        eng.set_option("debug_fail_wave", fail_wave)
        with pytest.raises(MemoryError, match="injected"):
            eng.calc_batch_device(d_a, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, stream=s1.cuda_stream)
        # no synchronisation here: the next call, on ANOTHER stream, must order itself behind whatever the failed call left running
        eng.calc_batch_device(d_b, 2 * n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, stream=s2.cuda_stream)
        s2.synchronize()
        got = d_flow.cpu().numpy()
        for i in range(n_pairs):
            np.testing.assert_array_equal(got[i], oracle.farneback(other[2 * i], other[2 * i + 1], levels=3, box_mode=oracle.BOX_BLOCKED))
    # host entry points: same injected failure, then a correct call
    a, b = frames[0], frames[1]
    with H.FarnebackEngine(w, h, 1, levels=3) as eng:
        eng.set_option("debug_fail_wave", 0)
        with pytest.raises(MemoryError):
            eng.calc(a, b)
        np.testing.assert_array_equal(eng.calc(a, b), oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_BLOCKED))
        eng.set_option("debug_fail_wave", 1)
        assert eng.stream_next(a) is None                      # wave 0 of the countdown
        with pytest.raises(MemoryError):
            eng.stream_next(b)
        assert not eng.stream_primed(w, h)                     # a failed turn drops the session
        assert eng.stream_next(a) is None
        np.testing.assert_array_equal(eng.stream_next(b), oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_BLOCKED))


@pytest.mark.parametrize("w,h,seed,kw", [c for c in CASES if c[2] in (10, 12, 13, 14, 15, 16, 17, 19, 20, 24, 25)] +
                         [(9, 7, 90, dict(levels=3)), (40, 1, 91, dict(levels=0, winsize=5, iterations=2)), (1, 40, 92, dict(levels=1))])
def test_pipeline_in_opencvs_literal_summation_order(H, oracle, w, h, seed, kw):
    """ "box_order" = 1: the box window of FarnebackUpdateFlow_Blur summed exactly as optflowgf.cpp sums it -- one double running
    sum per column-channel down the whole image with FLOAT row differences, one double running sum along each row
    (k_vsum_running + k_hsum_running_solve).  The result equals the oracle's OFO_BOX_RUNNING, the statement-by-statement order, BIT
    FOR BIT -- where the throughput kernels (restarted sums) agree with it to ~1e-6 px.  Pair call, batch and streaming turn."""
    if w > 8 and h > 8:
        prev, nxt, _ = translated_pair(h, w, seed, max_shift=5)
    else:
        rng = np.random.default_rng(seed)
        prev, nxt = rng.integers(0, 256, (h, w)).astype(np.uint8), rng.integers(0, 256, (h, w)).astype(np.uint8)
    ref = oracle.farneback(prev, nxt, box_mode=oracle.BOX_RUNNING, **kw)
    with H.FarnebackEngine(w, h, 2, **kw) as eng:
        eng.set_option("box_order", 1)
        np.testing.assert_array_equal(eng.calc(prev, nxt), ref)
        flow, _, _ = eng.calc_batch(np.stack([prev, nxt, nxt, prev]), want_danger=False)
        np.testing.assert_array_equal(flow[0], ref)
        np.testing.assert_array_equal(flow[1], oracle.farneback(nxt, prev, box_mode=oracle.BOX_RUNNING, **kw))
        assert eng.stream_next(prev) is None
        np.testing.assert_array_equal(eng.stream_next(nxt), ref)
        np.testing.assert_array_equal(eng.calc_reuse(nxt, prev), oracle.farneback(nxt, prev, box_mode=oracle.BOX_RUNNING, **kw))
        eng.set_option("box_order", 0)                      # and back to the throughput order on the same context
        np.testing.assert_array_equal(eng.calc(prev, nxt), oracle.farneback(prev, nxt, box_mode=oracle.BOX_BLOCKED, **kw))


def test_literal_order_1080p_config2_and_flags(H, oracle):
    """The literal order at BASELINE config 2's size (1920x1080, levels 5), on the warped family, with OPTFLOW_USE_INITIAL_FLOW, and
    with OPTFLOW_FARNEBACK_GAUSSIAN (whose window has one order only: the option changes nothing there)."""
    prev, nxt, _ = translated_pair(1080, 1920, 2001)
    with H.FarnebackEngine(1920, 1080, 1, levels=5) as eng:
        eng.set_option("box_order", 1)
        got = eng.calc(prev, nxt)
        np.testing.assert_array_equal(got, oracle.farneback(prev, nxt, levels=5, box_mode=oracle.BOX_RUNNING))
        fast = H.calculate_optical_flow(prev, nxt, levels=5)
        e = epe(got, fast)
        assert 0 < e.max() <= TOL_MAX_EPE and e.mean() <= TOL_MEAN_EPE          # the two orders differ, a little
    a, b, _, _ = warped_pair(251, 333, 5, zoom=1.04, angle_deg=2.0)
    init = (np.random.default_rng(3).standard_normal((251, 333, 2)) * 2).astype(np.float32)
    for kw, init_flow in ((dict(levels=3), None), (dict(levels=2, flags=4), init), (dict(levels=2, flags=256), None), (dict(levels=2, winsize=10), None)):
        with H.FarnebackEngine(333, 251, 1, **kw) as eng:
            eng.set_option("box_order", 1)
            got = eng.calc(a, b, init_flow.copy() if init_flow is not None else None)
            np.testing.assert_array_equal(got, oracle.farneback(a, b, box_mode=oracle.BOX_RUNNING, init_flow=init_flow, **kw))
