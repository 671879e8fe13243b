// ofarn_api_lk.hip -- C-ABI entry points of the sparse pyramidal Lucas-Kanade path and of the vector
// filter on given vectors (get_flow_lk, pathfinder_viewer.py:144-178).  Shared internals: ofarn_host.h.
#include "ofarn_host.h"

using namespace ofarn;
using namespace ofarn_host;

namespace {

int lk_levels(int w, int h, int win_w, int win_h, int max_level)   // buildOpticalFlowPyramid's stop rule
{
    int level = 0;
    while (level < max_level) {
        const int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= win_w || nh <= win_h) break;
        w = nw; h = nh; level++;
    }
    return level;
}

int check_lk_params(const ofarn_lk_params *p)
{
    if (!p) return fail(OFARN_E_INVALID, "lk params is NULL");
    if (p->max_level < 0) return fail(OFARN_E_INVALID, "maxLevel must be >= 0, got %d", p->max_level);
    if (p->win_w <= 2 || p->win_h <= 2) return fail(OFARN_E_INVALID, "winSize must be > 2 (cv2: CV_Assert), got %dx%d", p->win_w, p->win_h);
    if (p->win_w > 64 || p->win_h > 255) return fail(OFARN_E_UNSUPPORTED, "winSize up to 64 x 255 is built, got %dx%d", p->win_w, p->win_h);
    if (p->flags & ~(OFARN_LK_USE_INITIAL_FLOW | OFARN_LK_GET_MIN_EIGENVALS))
        return fail(OFARN_E_INVALID, "flags=%d: only OPTFLOW_USE_INITIAL_FLOW (4) and OPTFLOW_LK_GET_MIN_EIGENVALS (8) exist", p->flags);
    return OFARN_OK;
}

// (re)allocates the LK workspace for `frames` frames of w x h and `levels` pyramid levels above level 0
int ensure_lk_ws(ofarn_ctx *c, int w, int h, int frames, int levels)
{
    auto &K = c->lk;
    if (K.w == w && K.h == h && K.frames >= frames && K.levels >= levels) return OFARN_OK;
    for (uint8_t *p : K.pyr) if (p) (void)hipFree(p);
    for (int16_t *p : K.der) if (p) (void)hipFree(p);
    K = ofarn_ctx::LkWs();
    K.lw.assign(levels + 1, 0); K.lh.assign(levels + 1, 0);
    K.pyr.assign(levels + 1, nullptr); K.der.assign(levels + 1, nullptr);
    for (int l = 0; l <= levels; l++) {
        K.lw[l] = l == 0 ? w : (K.lw[l - 1] + 1) / 2;
        K.lh[l] = l == 0 ? h : (K.lh[l - 1] + 1) / 2;
        const size_t npx = (size_t)K.lw[l] * K.lh[l] * frames;
        if (l > 0 && hipMalloc((void **)&K.pyr[l], npx + 64) != hipSuccess) { (void)hipGetLastError(); return fail(OFARN_E_NOMEM, "LK pyramid does not fit"); }
        if (hipMalloc((void **)&K.der[l], npx * 2 * sizeof(int16_t) + 64) != hipSuccess) { (void)hipGetLastError(); return fail(OFARN_E_NOMEM, "LK derivatives do not fit"); }
    }
    K.w = w; K.h = h; K.frames = frames; K.levels = levels;
    return OFARN_OK;
}

// One wave of pairs: pyramid + derivatives of its frames, then the tracker level by level (coarse to fine).
int lk_wave(ofarn_ctx *c, hipStream_t s, const uint8_t *d_frames, int nframes, int npairs, int fstep, int i_off, int j_off,
            int w, int h, const float *d_pts, int npts, int pts_stride, const ofarn_lk_params &prm, float *d_next,
            uint8_t *d_status, float *d_err)
{
    const int levels = lk_levels(w, h, prm.win_w, prm.win_h, prm.max_level);
    int rc = ensure_lk_ws(c, w, h, nframes, levels);
    if (rc) return rc;
    auto &K = c->lk;
    for (int l = 0; l <= levels; l++) {
        const uint8_t *img = l == 0 ? d_frames : K.pyr[l];
        if (l > 0) launch_pyrdown_u8(s, l == 1 ? d_frames : K.pyr[l - 1], K.lw[l - 1], K.lh[l - 1], K.pyr[l], nframes);
        launch_scharr(s, img, K.lw[l], K.lh[l], K.der[l], i_off, fstep, npairs);   // derivatives of the first image of each pair only
    }
    int max_count = prm.max_count < 0 ? 0 : prm.max_count > 100 ? 100 : prm.max_count;
    double eps = prm.epsilon < 0 ? 0 : prm.epsilon > 10 ? 10 : prm.epsilon;
    for (int l = levels; l >= 0; l--) {
        LkLevelArgs A{};
        A.img = l == 0 ? d_frames : K.pyr[l];
        A.deriv = K.der[l];
        A.pts = d_pts; A.next_pts = d_next; A.status = d_status; A.err = d_err;
        A.w = K.lw[l]; A.h = K.lh[l]; A.npts = npts; A.pts_stride = pts_stride;
        A.fstep = fstep; A.i_off = i_off; A.j_off = j_off;
        A.win_w = prm.win_w; A.win_h = prm.win_h; A.level = l; A.top_level = levels; A.flags = prm.flags; A.max_count = max_count;
        A.scale = (float)(1. / (1 << l));
        A.min_eig = (float)prm.min_eig_threshold;
        A.eps2 = eps * eps;
        launch_lk_track(s, A, npairs);
    }
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

void ofarn_lk_default_params(ofarn_lk_params *p)
{
    if (!p) return;
    // cv2.calcOpticalFlowPyrLK defaults: winSize (21, 21), maxLevel 3, criteria (COUNT + EPS, 30, 0.01), flags 0, 1e-4
    p->win_w = 21; p->win_h = 21; p->max_level = 3; p->max_count = 30; p->epsilon = 0.01; p->flags = 0; p->min_eig_threshold = 1e-4;
}

int ofarn_lk_levels(const ofarn_lk_params *p, int w, int h)
{
    int rc = check_lk_params(p);
    if (rc) return rc;
    if (w < 1 || h < 1) return fail(OFARN_E_INVALID, "empty frame %dx%d", w, h);
    return lk_levels(w, h, p->win_w, p->win_h, p->max_level);
}

int ofarn_lk_calc_batch_device(ofarn_ctx *c, const uint8_t *d_frames, int n_frames, int w, int h, int pairs_mode, int reverse,
                               const float *d_pts, int npts, int pts_per_pair, const ofarn_lk_params *prm, float *d_next_pts,
                               uint8_t *d_status, float *d_err, void *hip_stream)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if ((rc = check_lk_params(prm))) return rc;
    if (!d_frames || !d_pts || !d_next_pts || !d_status || !d_err) return fail(OFARN_E_INVALID, "NULL argument");
    if (pairs_mode != OFARN_PAIRS_INDEPENDENT && pairs_mode != OFARN_PAIRS_CONSECUTIVE)
        return fail(OFARN_E_INVALID, "pairs_mode must be 0 or 1");
    const int n_pairs = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? n_frames - 1 : n_frames / 2;
    if (n_pairs < 0 || npts < 0 || (pairs_mode == OFARN_PAIRS_INDEPENDENT && (n_frames & 1)))
        return fail(OFARN_E_INVALID, "n_frames=%d does not form whole pairs in mode %d", n_frames, pairs_mode);
    if (n_pairs == 0 || npts == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    hipStream_t s = pick_stream(c, hip_stream);
    if ((rc = begin_call(c, s))) return rc;                       // the LK pyramid workspace is shared between calls
    const size_t fsz = (size_t)w * h;
    const int fstep = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? 1 : 2;
    const int wave = c->max_batch < 64 ? c->max_batch : 64;      // LK keeps a small workspace of its own
    // pair p = frames (a, a+1); cv2.calcOpticalFlowPyrLK(prev, next): track FROM prev TO next.  reverse tracks from the
    // later frame to the earlier one, as pathfinder_viewer.py:156 does (img2 -> img1).
    const int i_off = reverse ? 1 : 0, j_off = reverse ? 0 : 1;
    for (int p0 = 0; p0 < n_pairs; p0 += wave) {
        const int np = n_pairs - p0 < wave ? n_pairs - p0 : wave;
        const int nf = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? np + 1 : 2 * np;
        rc = lk_wave(c, s, d_frames + (size_t)p0 * fstep * fsz, nf, np, fstep, i_off, j_off, w, h,
                     d_pts + (pts_per_pair ? (size_t)p0 * npts * 2 : 0), npts, pts_per_pair ? npts : 0, *prm,
                     d_next_pts + (size_t)p0 * npts * 2, d_status + (size_t)p0 * npts, d_err + (size_t)p0 * npts);
        if (rc) return rc;
    }
    return end_call(c, s);
}

int ofarn_lk_calc(ofarn_ctx *c, const uint8_t *h_prev, const uint8_t *h_next, int w, int h, int stride, const float *h_pts,
                  int npts, const ofarn_lk_params *prm, float *h_next_pts, uint8_t *h_status, float *h_err)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if ((rc = check_lk_params(prm))) return rc;
    if (!h_prev || !h_next || !h_pts || !h_next_pts || !h_status || !h_err) return fail(OFARN_E_INVALID, "NULL argument");
    if (stride < w) return fail(OFARN_E_INVALID, "stride %d < width %d", stride, w);
    if (npts < 0) return fail(OFARN_E_INVALID, "npts < 0");
    if (npts == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    const size_t fsz = (size_t)w * h;
    if ((rc = ensure_staging(c, 2 * fsz, 0, 0))) return rc;
    DevTmp pts, nxt, st, er;
    if ((rc = pts.alloc((size_t)npts * 8)) || (rc = nxt.alloc((size_t)npts * 8)) || (rc = st.alloc(npts)) || (rc = er.alloc((size_t)npts * 4)))
        return rc;
    if ((rc = begin_call(c, c->stream))) return rc;
    HIP_TRY(hipMemcpy2DAsync(c->st_frames, w, h_prev, stride, w, h, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpy2DAsync(c->st_frames + fsz, w, h_next, stride, w, h, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(pts.p, h_pts, (size_t)npts * 8, hipMemcpyHostToDevice, c->stream));
    if (prm->flags & OFARN_LK_USE_INITIAL_FLOW)
        HIP_TRY(hipMemcpyAsync(nxt.p, h_next_pts, (size_t)npts * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = lk_wave(c, c->stream, c->st_frames, 2, 1, 2, 0, 1, w, h, pts.as<float>(), npts, 0, *prm, nxt.as<float>(),
                      st.as<uint8_t>(), er.as<float>())))
        return rc;
    HIP_TRY(hipMemcpyAsync(h_next_pts, nxt.p, (size_t)npts * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_status, st.p, npts, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_err, er.p, (size_t)npts * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return end_call(c, c->stream);
}

int ofarn_vector_filter_device(ofarn_ctx *c, const float *d_vecs, int n, int w, int h, uint8_t *d_mask, uint8_t *d_v,
                               int32_t *d_iflow, void *hip_stream)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!d_vecs || !d_mask || !d_v) return fail(OFARN_E_INVALID, "vectors, mask and v must not be NULL");
    if (n < 0) return fail(OFARN_E_INVALID, "n < 0");
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    if (c->P == 0) return OFARN_OK;
    hipStream_t s = pick_stream(c, hip_stream);
    launch_grid_filter(s, nullptr, w, h, n, c->d_pts, c->P, c->prm.filter_variant, d_mask, d_v, d_iflow, d_vecs);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_vector_filter(ofarn_ctx *c, const float *h_vecs, int n, int w, int h, uint8_t *h_mask, uint8_t *h_v, int32_t *h_iflow)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_vecs || !h_mask || !h_v) return fail(OFARN_E_INVALID, "vectors, mask and v must not be NULL");
    if (n < 0) return fail(OFARN_E_INVALID, "n < 0");
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    if (c->P == 0) return OFARN_OK;
    const size_t P = (size_t)c->P;
    DevTmp vec, mk, vv, ifl;
    if ((rc = vec.alloc(P * n * 8)) || (rc = mk.alloc(P * n)) || (rc = vv.alloc(P * n)) || (rc = ifl.alloc(P * n * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(vec.p, h_vecs, P * n * 8, hipMemcpyHostToDevice, c->stream));
    launch_grid_filter(c->stream, nullptr, w, h, n, c->d_pts, c->P, c->prm.filter_variant, mk.as<uint8_t>(), vv.as<uint8_t>(),
                       h_iflow ? ifl.as<int32_t>() : nullptr, vec.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_mask, mk.p, P * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_v, vv.p, P * n, hipMemcpyDeviceToHost, c->stream));
    if (h_iflow) HIP_TRY(hipMemcpyAsync(h_iflow, ifl.p, P * n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OFARN_OK;
}

int ofarn_draw_vectors_device(ofarn_ctx *c, const int32_t *d_iflow, const uint8_t *d_mask, int n, int w, int h, int draw_bad,
                              uint8_t *d_out, void *hip_stream)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!d_out) return fail(OFARN_E_INVALID, "out is NULL");
    if (n < 0) return fail(OFARN_E_INVALID, "n < 0");
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    if (c->P > 0 && (!d_iflow || !d_mask)) return fail(OFARN_E_INVALID, "iflow and mask must not be NULL");
    hipStream_t s = pick_stream(c, hip_stream);
    HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)n * w * h * 3, s));
    if (c->P > 0) launch_draw_vectors(s, c->d_pts, d_iflow, d_mask, c->P, n, w, h, draw_bad, d_out);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_draw_vectors(ofarn_ctx *c, const int32_t *h_iflow, const uint8_t *h_mask, int n, int w, int h, int draw_bad, uint8_t *h_out)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_out) return fail(OFARN_E_INVALID, "out is NULL");
    if (n < 0) return fail(OFARN_E_INVALID, "n < 0");
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    const size_t P = (size_t)c->P, img = (size_t)w * h * 3;
    if (P > 0 && (!h_iflow || !h_mask)) return fail(OFARN_E_INVALID, "iflow and mask must not be NULL");
    DevTmp ifl, mk, out;
    if ((rc = ifl.alloc(P * 8 + 8)) || (rc = mk.alloc(P + 8)) || (rc = out.alloc(img))) return rc;
    for (int i = 0; i < n; i++) {
        if (P > 0) {
            HIP_TRY(hipMemcpyAsync(ifl.p, h_iflow + (size_t)i * P * 2, P * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(mk.p, h_mask + (size_t)i * P, P, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipMemsetAsync(out.p, 0, img, c->stream));
        if (P > 0) launch_draw_vectors(c->stream, c->d_pts, ifl.as<int32_t>(), mk.as<uint8_t>(), c->P, 1, w, h, draw_bad, out.as<uint8_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_out + (size_t)i * img, out.p, img, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return OFARN_OK;
}

#pragma GCC visibility pop
}  // extern "C"
