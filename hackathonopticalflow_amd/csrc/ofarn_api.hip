// ofarn_api.hip -- host side of libofarn.so: the core of the C-ABI of include/ofarn.h (context, plan, the dense
// Farneback wave schedule, batch entry points, grid filter).  The front end / visualiser / single-stage entry points
// are in ofarn_api_extras.hip, sparse LK in ofarn_api_lk.hip; shared internals in ofarn_host.h.
//
// Owns the HBM workspace, the per-frame-size level plan (optflowgf.cpp calc(): level sizes,
// Gaussian kernels, resize tables) and the per-level launch schedule
//     A (level image) -> B (poly expansion) -> E (flow upsample) -> C, [D, C] x (I-1), D
// for a wave of frame pairs at a time.  No oracle or CPU fallback exists on this path: if a HIP
// call fails the entry point returns OFARN_E_HIP.
#include "ofarn_host.h"

using namespace ofarn;
using namespace ofarn_host;

namespace ofarn_host {

thread_local std::string g_err;
int g_fake_current_device = -1;
thread_local int t_last_restored_device = -1, t_device_scopes = 0;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}


inline int cv_round(double v) { return (int)lrint(v); }   // cvRound: half to even
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }

// getGaussianKernel(n, sigma, CV_32F) -- smooth.dispatch.cpp
std::vector<float> gaussian_kernel(int n, double sigma)
{
    std::vector<float> out(n);
    if ((n & 1) && n <= 7 && sigma <= 0) {
        static const float tab[4][7] = {{1.f},
                                        {0.25f, 0.5f, 0.25f},
                                        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
        for (int i = 0; i < n; i++) out[i] = tab[n >> 1][i];
        return out;
    }
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    std::vector<double> v(n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        v[i] = std::exp(scale2X * x * x);
        sum += v[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) out[i] = (float)(v[i] * sum);
    return out;
}

// resize(INTER_LINEAR) coordinate tables -- resize.cpp
void resize_tables(int ssize, int dsize, std::vector<int> &ofs, std::vector<float> &alpha)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    ofs.resize(dsize);
    alpha.resize(dsize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        alpha[d] = f;
    }
}

// FarnebackPrepareGaussian -- optflowgf.cpp
bool poly_prepare(int n, double sigma, PolyCoef &c)
{
    if (n < 1 || n > kMaxPolyN) return false;
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    std::vector<float> gb(2 * n + 1), xgb(2 * n + 1), xxgb(2 * n + 1);
    float *g = gb.data() + n, *xg = xgb.data() + n, *xxg = xxgb.data() + n;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    // 6x6 inverse, Gauss-Jordan with partial pivoting
    double a[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { a[i][j] = G[i][j]; a[i][j + 6] = (i == j); }
    for (int col = 0; col < 6; col++) {
        int p = col;
        for (int r = col + 1; r < 6; r++) if (std::fabs(a[r][col]) > std::fabs(a[p][col])) p = r;
        if (p != col) for (int j = 0; j < 12; j++) std::swap(a[col][j], a[p][j]);
        const double d = 1. / a[col][col];
        for (int j = 0; j < 12; j++) a[col][j] *= d;
        for (int r = 0; r < 6; r++) if (r != col) {
            const double f = a[r][col];
            if (f != 0) for (int j = 0; j < 12; j++) a[r][j] -= f * a[col][j];
        }
    }
    memset(&c, 0, sizeof(c));
    c.n = n;
    for (int k = 0; k <= n; k++) { c.g[k] = g[k]; c.xg[k] = xg[k]; c.xxg[k] = xxg[k]; }
    c.ig11 = a[1][7]; c.ig03 = a[0][9]; c.ig33 = a[3][9]; c.ig55 = a[5][11];
    return true;
}

// computeResizeAreaTab -- resize.cpp: the source cells [dx*scale, (dx+1)*scale) with fractional end weights
struct AreaAxis {
    std::vector<int> start, si;
    std::vector<float> alpha;
};
void area_axis(int ssize, int dsize, double scale, AreaAxis &t)
{
    t.start.assign(1, 0);
    t.si.clear();
    t.alpha.clear();
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) { t.si.push_back(sx1 - 1); t.alpha.push_back((float)((sx1 - fsx1) / cell)); }
        for (int sx = sx1; sx < sx2; sx++) { t.si.push_back(sx); t.alpha.push_back((float)(1.0 / cell)); }
        if (fsx2 - sx2 > 1e-3) { t.si.push_back(sx2); t.alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)); }
        t.start.push_back((int)t.si.size());
    }
}


int axis_points(int size, int step, std::vector<int> *out)
{
    // pathfinder_viewer.py:255-262 + np.mgrid[indent:size:step].astype(int)
    const double indent = ((size / step) % 2 == 1) ? (size % step) / 2.0 : ((size % step) + step) / 2.0;
    int n = (int)std::ceil((size - indent) / (step * 1.0));
    if (n < 0) n = 0;
    if (out) {
        out->resize(n);
        for (int i = 0; i < n; i++) (*out)[i] = (int)(i * (double)step + indent);
    }
    return n;
}

int crop_levels(int W, int H, double pyr_scale, int levels)
{
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (W * scale < 32 || H * scale < 32) break;   // min_size = 32
    }
    return k;
}

void level_geom(int W, int H, double pyr_scale, int k, int &w, int &h, double &sigma, int &ksize)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    sigma = (1. / scale - 1) * 0.5;
    ksize = cv_round(sigma * 5) | 1;
    if (ksize < 3) ksize = 3;
    w = cv_round(W * scale);
    h = cv_round(H * scale);
}

int check_params(const ofarn_params *p)
{
    if (!p) return fail(OFARN_E_INVALID, "params is NULL");
    if (!(p->pyr_scale < 1) || !(p->pyr_scale > 0))
        return fail(OFARN_E_INVALID, "pyr_scale must be in (0, 1) (cv2: CV_Assert(pyrScale_ < 1)), got %g", p->pyr_scale);
    if (p->levels < 0) return fail(OFARN_E_INVALID, "levels must be >= 0, got %d", p->levels);
    if (p->winsize < 2 || p->winsize > blur_solve_max_winsize())
        return fail(OFARN_E_INVALID, "winsize must be in [2, %d], got %d", blur_solve_max_winsize(), p->winsize);
    if (p->iterations < 0) return fail(OFARN_E_INVALID, "iterations must be >= 0, got %d", p->iterations);
    if (p->poly_n < 1 || p->poly_n > kMaxPolyN)
        return fail(OFARN_E_INVALID, "poly_n must be in [1, %d], got %d", kMaxPolyN, p->poly_n);
    if (p->flags & ~(OFARN_FLAG_FARNEBACK_GAUSSIAN | OFARN_FLAG_USE_INITIAL_FLOW))
        return fail(OFARN_E_UNSUPPORTED, "flags=%d: only OPTFLOW_USE_INITIAL_FLOW (4) and OPTFLOW_FARNEBACK_GAUSSIAN (256) exist", p->flags);
    if ((p->flags & OFARN_FLAG_FARNEBACK_GAUSSIAN) && p->winsize / 2 > 60)
        return fail(OFARN_E_INVALID, "winsize must be <= 121 with OPTFLOW_FARNEBACK_GAUSSIAN, got %d", p->winsize);
    if (p->grid_step < 1) return fail(OFARN_E_INVALID, "grid_step must be >= 1, got %d", p->grid_step);
    if (p->filter_variant != 0 && p->filter_variant != 1)
        return fail(OFARN_E_INVALID, "filter_variant must be 0 or 1, got %d", p->filter_variant);
    return OFARN_OK;
}

}  // namespace ofarn_host


namespace ofarn_host {

void free_plan(ofarn_ctx *c)
{
    for (void *p : c->plan_allocs) (void)hipFree(p);
    c->plan_allocs.clear();
    c->lv.clear();
    c->d_pts = nullptr;
    c->plan_w = c->plan_h = 0;
    c->P = 0;
}

template <typename T>
int upload(ofarn_ctx *c, const std::vector<T> &v, T **out)
{
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, v.size() * sizeof(T) + 16));
    c->plan_allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<T *>(d);
    return OFARN_OK;
}

// Device tables for resize(INTER_AREA) from sw x sh down to dw x dh (allocations owned by the plan).
int build_area_tab(ofarn_ctx *c, int sw, int sh, int dw, int dh, AreaTabHost &out)
{
    if (dw > sw || dh > sh) return fail(OFARN_E_INVALID, "INTER_AREA is only built for shrinking (%dx%d -> %dx%d)", sw, sh, dw, dh);
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    const int isx = cv_round(scale_x), isy = cv_round(scale_y);      // saturate_cast<int>(double)
    out = AreaTabHost();
    out.fast = std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON;
    out.iscale_x = isx;
    out.iscale_y = isy;
    if (out.fast) return OFARN_OK;
    AreaAxis ax, ay;
    area_axis(sw, dw, scale_x, ax);
    area_axis(sh, dh, scale_y, ay);
    int rc;
    if ((rc = upload(c, ax.start, &out.xstart)) || (rc = upload(c, ax.si, &out.xsi)) || (rc = upload(c, ax.alpha, &out.xalpha)) ||
        (rc = upload(c, ay.start, &out.ystart)) || (rc = upload(c, ay.si, &out.ysi)) || (rc = upload(c, ay.alpha, &out.yalpha)))
        return rc;
    return OFARN_OK;
}

int make_plan(ofarn_ctx *c, int w, int h)
{
    if (c->plan_w == w && c->plan_h == h) return OFARN_OK;
    free_plan(c);
    const int nlev = crop_levels(w, h, c->prm.pyr_scale, c->prm.levels);
    c->lv.resize(nlev + 1);
    for (int k = 0; k <= nlev; k++) {
        Level &L = c->lv[k];
        level_geom(w, h, c->prm.pyr_scale, k, L.w, L.h, L.sigma, L.ksize);
        if (L.w < 1 || L.h < 1) return fail(OFARN_E_INVALID, "level %d is empty", k);
        int rc;
        {
            const std::vector<float> kv = gaussian_kernel(L.ksize, L.sigma);
            if (L.ksize == 3) for (int i = 0; i < 3; i++) L.h_kern3[i] = kv[i];
            L.h_kern = kv;
            if ((rc = upload(c, kv, &L.d_kern))) return rc;
        }
        std::vector<int> ofs;
        std::vector<float> al;
        resize_tables(w, L.w, ofs, al);
        if ((rc = upload(c, ofs, &L.d_xofs)) || (rc = upload(c, al, &L.d_xa))) return rc;
        resize_tables(h, L.h, ofs, al);
        if ((rc = upload(c, ofs, &L.d_yofs)) || (rc = upload(c, al, &L.d_ya))) return rc;
    }
    for (int k = 0; k < nlev; k++) {
        Level &L = c->lv[k];
        const Level &S = c->lv[k + 1];
        std::vector<int> ofs;
        std::vector<float> al;
        int rc;
        resize_tables(S.w, L.w, ofs, al);
        if ((rc = upload(c, ofs, &L.d_fxofs)) || (rc = upload(c, al, &L.d_fxa))) return rc;
        resize_tables(S.h, L.h, ofs, al);
        if ((rc = upload(c, ofs, &L.d_fyofs)) || (rc = upload(c, al, &L.d_fya))) return rc;
    }
    c->area = AreaTabHost();
    c->init_scale = 1.f;
    if ((c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) && nlev > 0) {
        // resize(flow0, flow, Size(width, height), 0, 0, INTER_AREA); flow *= scale  (optflowgf.cpp calc(), coarsest level)
        int rc;
        if ((rc = build_area_tab(c, w, h, c->lv[nlev].w, c->lv[nlev].h, c->area))) return rc;
        double scale = 1;
        for (int i = 0; i < nlev; i++) scale *= c->prm.pyr_scale;
        c->init_scale = (float)scale;
    }
    std::vector<int> xs, ys;
    axis_points(w, c->prm.grid_step, &xs);
    axis_points(h, c->prm.grid_step, &ys);
    std::vector<int> pts;
    for (int x : xs)
        for (int y : ys) { pts.push_back(x); pts.push_back(y); }
    c->P = (int)(pts.size() / 2);
    if (c->P > 0) {
        int rc;
        if ((rc = upload(c, pts, &c->d_pts))) return rc;
    }
    c->plan_w = w;
    c->plan_h = h;
    return OFARN_OK;
}

hipEvent_t prof_event(ofarn_ctx *c)
{
    if (!c->prof_free.empty()) { hipEvent_t e = c->prof_free.back(); c->prof_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}


// Which kernel builds each level image of a wave and what the row-pass intermediates need (offsets only; nothing is launched).
struct WavePlan {
    size_t tmp_off[32] = {0};
    bool has_tmp[32] = {false}, hdirect[32] = {false};
    int hl_level[12] = {0};
    HLevels HL{};
    size_t tmp_need = 0, I_need = 0;     // floats
    bool multi = false;
};

// The marching stage-A kernels (k_level_direct, k_level_hdirect) give each thread a long strip; they win once a wave holds
// enough pixels to fill the chip that way: direct_min_frames counts frames of 1920 x 1080 (16: 4K from 4 frames on).
static inline bool direct_wave(const ofarn_ctx *c, int nframes, int w, int h)
{
    return (double)nframes * w * h >= (double)c->direct_min_frames * (1920.0 * 1080.0);
}

void plan_wave(const ofarn_ctx *c, const uint8_t *d_frames, int nframes, int w, int h, WavePlan &p)
{
    const int nlev = (int)c->lv.size() - 1;
    const bool march0 = !c->force_generic && polyexp_march_supported(c->prm.poly_n);
    p.HL.symm = c->row_small_symm;
    size_t off = 0, single = 0;
    bool ok = !c->force_generic;
    for (int k = nlev; k >= 0; k--) {
        const Level &L = c->lv[k];
        if (march0 && L.w == w && L.h == h && L.ksize == 3) continue;   // fused into the poly expansion, no level image
        p.I_need = std::max(p.I_need, (size_t)nframes * L.w * L.h);
        if (!c->force_generic && direct_wave(c, nframes, w, h) && level_direct_supported(d_frames, w, h, L.w, L.h, L.ksize))
            continue;   // built by k_level_direct, no tmp
        const size_t need = (size_t)nframes * h * L.w * 2;
        single = std::max(single, need);
        if (!ok) continue;
        if (direct_wave(c, nframes, w, h) && level_hdirect_supported(d_frames, w, L.w, L.ksize))
            p.hdirect[k] = true;   // 1/16, 1/32, 1/64 widths: row pass straight from the frames, no LDS staging
        else if (p.HL.n >= 12) { ok = false; continue; }
        else {
            p.hl_level[p.HL.n] = k;
            p.HL.lv[p.HL.n++] = HLevel{L.d_kern, L.d_xofs, nullptr, L.w, L.ksize};
            if (L.ksize / 2 > p.HL.rmax) p.HL.rmax = L.ksize / 2;
        }
        p.has_tmp[k] = true;
        p.tmp_off[k] = off;
        off += need;
    }
    p.HL.rmax = (p.HL.rmax + 3) & ~3;   // border width in LDS: a multiple of 4 keeps the 16-byte staging writes aligned
    p.multi = ok && (p.HL.n == 0 || hpass_multi_lds_bytes(w, p.HL.rmax) <= 60 * 1024);
    p.tmp_need = p.multi ? off : single;
}

// One wave: npairs <= max_batch pairs, frames already in HBM.
// d_init (OPTFLOW_USE_INITIAL_FLOW): full-resolution start flows float[npairs][h][w][2]; may alias d_flow.
int run_wave(ofarn_ctx *c, hipStream_t s, const uint8_t *d_frames, int npairs, int pairs_mode, int w,
             int h, float *d_flow, uint8_t *d_mask, uint8_t *d_v, int wi, const float *d_init, ofarn_ctx::Stream *st, bool st_pair)
{
    ofarn_ctx::Workspace &ws = c->ws[wi];
    // streaming turn: one new frame into slot `snew`; the pair is (slot cur, slot snew), one pair
    // st_pair (the session holds nothing): d_frames are BOTH frames of a pair; they go to slots 0 and 1 in one set of launches and
    // are iterated at once -- the session afterwards holds the second frame in slot 1
    const int snew = st ? ((st->have && !st_pair) ? st->cur ^ 1 : 0) : 0;
    const bool iterate = !st || st->have || st_pair;
    if (st) npairs = 1;
    const int fstep = st ? (st_pair ? 1 : (snew > st->cur ? 1 : -1)) : (pairs_mode == OFARN_PAIRS_CONSECUTIVE ? 1 : 2);
    const int nframes = st ? (st_pair ? 2 : 1) : (pairs_mode == OFARN_PAIRS_CONSECUTIVE ? npairs + 1 : 2 * npairs);
    // where level k's polynomial expansions are written (stages A + B) and read (iterations)
    auto R_ab = [&](int k) -> float * {
        return st ? st->R + st->off[k] + (size_t)snew * r_frame_stride((size_t)c->lv[k].w * c->lv[k].h) : ws.R;
    };
    auto R_it = [&](int k) -> const float * {
        return st ? st->R + st->off[k] + (size_t)(st_pair ? 0 : st->cur) * r_frame_stride((size_t)c->lv[k].w * c->lv[k].h) : ws.R;
    };
    const size_t fsz = (size_t)w * h;
    const int nlev = (int)c->lv.size() - 1;
    float *prev = nullptr;
    int pw = 0, ph = 0;
    const bool gauss = (c->prm.flags & OFARN_FLAG_FARNEBACK_GAUSSIAN) != 0;
    const bool running = c->box_running && !gauss;      // the Gaussian window has no running sums: one order only
    const bool fused = !c->force_generic && !running && c->prm.iterations >= 1 &&
                       (gauss ? flow_iter_gauss_supported(c->prm.winsize) : flow_iter_supported(c->prm.winsize));
    // Row pass of the level build for all levels that need one, in a single launch; tmp_of[k] is where level k's
    // rows go.  The plan is made first (offsets only), then the workspace is grown to what it needs.
    WavePlan wp;
    plan_wave(c, d_frames, nframes, w, h, wp);
    float *tmp_of[32] = {nullptr};
    const bool (&has_tmp)[32] = wp.has_tmp;
    const bool (&hdirect)[32] = wp.hdirect;
    HLevels &HL = wp.HL;
    const bool multi = wp.multi;
    {
        // the fused iteration kernel keeps M on chip; the literal-order mode also needs its double column sums (10 floats per pixel)
        const size_t M_need = (fused || !iterate) ? 0 : (size_t)npairs * fsz * (running ? 15 : 5);
        int rc = ws_reserve(c, wi, wp.tmp_need, wp.I_need, 0, M_need, 0, stream_is_capturing(s));
        if (rc) return rc;
    }
    if (c->debug_fail_wave >= 0 && c->debug_fail_wave-- == 0)      // test hook (ofarn_set_option "debug_fail_wave")
        return fail(OFARN_E_NOMEM, "injected failure (debug_fail_wave)");
    // Streaming turn, latency mode: the new frame's level images and polynomial expansions (12 small launches) do not depend on
    // the iterations of the coarser levels, only the other way round.  They go to an internal stream in level order, an event
    // behind each level's expansion; the caller's stream runs the iteration chain and waits for a level's event just before
    // that level's first iteration.  The critical path is then stages A + B of the COARSEST level + the iterations, not all of
    // A + B.  Possible here because every level has its own R slots (a batch wave reuses one R buffer level after level);
    // ws.tmp / ws.I are only touched by the A + B chain, which stays on one stream.  Per-kernel timing keeps one stream.
    const bool overlap = st && iterate && !c->prof_on && c->stream_overlap && c->aux[0] && c->ev_fork;
    hipStream_t sab = overlap ? c->aux[0] : s;
    if (overlap) {
        for (int k = 0; k <= nlev; k++)
            if (!c->ev_level[k] && hipEventCreateWithFlags(&c->ev_level[k], hipEventDisableTiming) != hipSuccess)
                return fail(OFARN_E_HIP, "event creation failed");
        HIP_TRY(hipEventRecord(c->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(sab, c->ev_fork, 0));
    }
    if (multi) {
        for (int k = 0; k <= nlev; k++) if (has_tmp[k]) tmp_of[k] = ws.tmp + wp.tmp_off[k];
        for (int i = 0; i < HL.n; i++) HL.lv[i].dst = tmp_of[wp.hl_level[i]];
        if (HL.n > 0) {
            double units = 0;
            for (int i = 0; i < HL.n; i++) units += (double)HL.lv[i].dw * h * nframes;
            timed(c, sab, OFARN_STAGE_LEVEL_H, 31, units, [&] { launch_level_hpass_multi(sab, d_frames, fsz, w, h, nframes, HL); });
        }
        for (int k = nlev; k >= 0; k--)
            if (hdirect[k]) {
                const Level &L = c->lv[k];
                timed(c, sab, OFARN_STAGE_LEVEL_H, k, (double)L.w * h * nframes, [&] {
                    launch_level_hdirect(sab, d_frames, fsz, w, h, nframes, L.h_kern.data(), L.ksize, tmp_of[k], L.w);
                });
            }
    }
    // OPTFLOW_USE_INITIAL_FLOW: the coarsest level starts from resize(flow0, INTER_AREA) * scale instead of zero
    const float *init_cur = nullptr;
    if (d_init && iterate) {
        const Level &Lc = c->lv[nlev];
        if (nlev == 0)   // same size: resize() copies, scale = 1
            HIP_TRY(hipMemcpyAsync(ws.flowA, d_init, fsz * 2 * sizeof(float) * npairs, hipMemcpyDeviceToDevice, s));
        else
            timed(c, s, OFARN_STAGE_INIT_FLOW, nlev, (double)Lc.w * Lc.h * npairs, [&] {
                launch_resize_area(s, d_init, w, h, ws.flowA, Lc.w, Lc.h, npairs, c->area, c->init_scale);
            });
        init_cur = ws.flowA;
    }
    // With the overlap the level loop runs twice: first every level's stages A + B go to the internal stream (so that stream
    // can run ahead however the two streams share hardware queues), then the iteration chain to the caller's.  sync_at[k]: the
    // chain waits for an event recorded behind level k's expansion ("stream_overlap" 1: behind every level's; 2: only behind the
    // coarsest level's, then behind every second one's -- a wait for work that finished long ago costs the chain less than one
    // for work that finishes just then, and by the time the chain has iterated a level the other stream is two levels ahead).
    bool sync_at[32];
    for (int k = 0; k <= nlev; k++) sync_at[k] = c->stream_overlap < 2 || k == nlev || k == 0 || ((nlev - k) % 2 == 1);
    int waited = nlev + 1;           // the chain has waited for the event behind this level (and so for every level above it)
    const int npass = overlap ? 2 : 1;
    for (int pass = 0; pass < npass; pass++)
    for (int k = nlev; k >= 0; k--) {
        const Level &L = c->lv[k];
        const size_t npx = (size_t)L.w * L.h;
        const double upx = (double)npx * npairs, ufr = (double)npx * nframes;
        const float mul = (float)(1. / c->prm.pyr_scale);
        // stages A + B: level image and polynomial expansion of every frame of the wave
        const bool march = !c->force_generic && polyexp_march_supported(c->prm.poly_n);
        const bool lds_ok = (size_t)(w + 2 * (L.ksize / 2)) * 4 * 33 / 32 + 4 * (size_t)L.ksize + 64 <= 60 * 1024;
        if (overlap && pass == 1) {
            // the chain: wait for the event that covers this level -- the nearest one recorded at or below it
            if (iterate && waited > k) {
                int e = k;
                while (!sync_at[e]) e--;             // sync_at[0] is always set
                HIP_TRY(hipStreamWaitEvent(s, c->ev_level[e], 0));
                waited = e;
            }
        } else if (march && L.w == w && L.h == h && L.ksize == 3) {
            // scale 1: 3-tap blur fused into the polynomial expansion, frames read directly
            timed(c, sab, OFARN_STAGE_POLYEXP, k, ufr, [&] {
                launch_polyexp_march(sab, d_frames, fsz, 1, R_ab(k), L.w, L.h, nframes, c->poly, L.h_kern3);
            });
        } else if (!c->force_generic && direct_wave(c, nframes, w, h) &&
                   level_direct_supported(d_frames, w, h, L.w, L.h, L.ksize)) {
            // exact 1/2, 1/4, 1/8 levels: row pass + column pass + resize in one kernel straight from the frames
            timed(c, sab, OFARN_STAGE_LEVEL_V, k, ufr, [&] {
                launch_level_direct(sab, d_frames, fsz, w, h, nframes, L.h_kern.data(), L.ksize, ws.I, L.w, L.h, c->row_small_symm);
            });
            timed(c, sab, OFARN_STAGE_POLYEXP, k, ufr, [&] {
                if (march) launch_polyexp_march(sab, ws.I, npx, 0, R_ab(k), L.w, L.h, nframes, c->poly, L.h_kern3);
                else launch_polyexp(sab, ws.I, R_ab(k), L.w, L.h, nframes, c->poly);
            });
        } else {
            float *tmpk = tmp_of[k] ? tmp_of[k] : ws.tmp;
            if (!tmp_of[k])
                timed(c, sab, OFARN_STAGE_LEVEL_H, k, ufr, [&] {
                    if (!c->force_generic && lds_ok)
                        launch_level_hpass_lds(sab, d_frames, fsz, w, h, nframes, L.d_kern, L.ksize, L.d_xofs, L.w, ws.tmp, c->row_small_symm);
                    else
                        launch_level_hpass(sab, d_frames, fsz, w, h, nframes, L.d_kern, L.ksize, L.d_xofs, L.w, ws.tmp, c->row_small_symm);
                });
            timed(c, sab, OFARN_STAGE_LEVEL_V, k, ufr, [&] {
                launch_level_vpass(sab, tmpk, h, L.w, L.h, nframes, L.d_kern, L.ksize, L.d_xa, L.d_yofs, L.d_ya, ws.I);
            });
            timed(c, sab, OFARN_STAGE_POLYEXP, k, ufr, [&] {
                if (march) launch_polyexp_march(sab, ws.I, npx, 0, R_ab(k), L.w, L.h, nframes, c->poly, L.h_kern3);
                else launch_polyexp(sab, ws.I, R_ab(k), L.w, L.h, nframes, c->poly);
            });
        }
        if (!iterate) continue;      // streaming, first frame: nothing to pair it with yet
        if (overlap && pass == 0) {
            if (sync_at[k]) HIP_TRY(hipEventRecord(c->ev_level[k], sab));
            continue;
        }
        if (fused) {
            // stages (E +) C + D fused per iteration; flow ping-pongs between two buffers, the last
            // iteration of level 0 writes the caller's buffer.  The coarse flow is only read by
            // iteration 0, so its buffer is free again from iteration 1 on.
            const float *cur = k == nlev ? init_cur : nullptr;
            for (int i = 0; i < c->prm.iterations; i++) {
                const float *busy = (i == 0 && prev) ? prev : cur;
                float *out = (i == c->prm.iterations - 1 && k == 0 && d_flow) ? d_flow
                             : (busy == ws.flowA ? ws.flowB : ws.flowA);
                const int mode = i == 0 ? (prev ? 1 : (cur ? 2 : 0)) : 2;
                timed(c, s, OFARN_STAGE_FLOW_ITER, k, upx, [&] {
                    if (gauss)
                        launch_flow_iter_gauss(s, R_it(k), fstep, cur, out, L.w, L.h, npairs, c->prm.winsize, c->h_gwin.data(), mode,
                                               prev, pw, ph, L.d_fxofs, L.d_fxa, L.d_fyofs, L.d_fya, mul);
                    else
                        launch_flow_iter(s, R_it(k), fstep, cur, out, L.w, L.h, npairs, c->prm.winsize, mode, prev, pw, ph,
                                         L.d_fxofs, L.d_fxa, L.d_fyofs, L.d_fya, mul, c->tile_mode);
                });
                cur = out;
            }
            prev = const_cast<float *>(cur); pw = L.w; ph = L.h;
            continue;
        }
        float *flow = (k == 0 && d_flow) ? d_flow : (prev == ws.flowA ? ws.flowB : ws.flowA);
        // literal-order mode: the double column sums V sit at the (256-byte aligned) start of the M buffer, M behind them
        double *Vk = running ? reinterpret_cast<double *>(ws.M) : nullptr;
        float *Mk = running ? ws.M + (size_t)npairs * npx * 10 : ws.M;
        if (!prev && init_cur) {
            if (flow != init_cur)
                HIP_TRY(hipMemcpyAsync(flow, init_cur, npx * 2 * sizeof(float) * npairs, hipMemcpyDeviceToDevice, s));
        } else if (!prev) HIP_TRY(hipMemsetAsync(flow, 0, npx * 2 * sizeof(float) * npairs, s));
        else
            timed(c, s, OFARN_STAGE_UPSAMPLE, k, upx, [&] {
                launch_flow_upsample(s, prev, pw, ph, flow, L.w, L.h, npairs, L.d_fxofs, L.d_fxa, L.d_fyofs,
                                     L.d_fya, mul);
            });
        timed(c, s, OFARN_STAGE_MATRICES, k, upx, [&] { launch_update_matrices(s, R_it(k), fstep, flow, Mk, L.w, L.h, npairs); });
        for (int i = 0; i < c->prm.iterations; i++) {
            timed(c, s, OFARN_STAGE_BLUR_SOLVE, k, upx, [&] {
                if (gauss) launch_gauss_solve(s, Mk, flow, L.w, L.h, npairs, c->prm.winsize, c->d_gwin);
                else if (running) launch_blur_solve_running(s, Mk, Vk, flow, L.w, L.h, npairs, c->prm.winsize);
                else launch_blur_solve(s, Mk, flow, L.w, L.h, npairs, c->prm.winsize);
            });
            if (i < c->prm.iterations - 1)
                timed(c, s, OFARN_STAGE_MATRICES, k, upx, [&] { launch_update_matrices(s, R_it(k), fstep, flow, Mk, L.w, L.h, npairs); });
        }
        prev = flow; pw = L.w; ph = L.h;
    }
    if (iterate && (d_mask || d_v) && c->P > 0) {
        if (!d_mask || !d_v) return fail(OFARN_E_INVALID, "danger mask and v must be given together");
        timed(c, s, OFARN_STAGE_GRID_FILTER, 0, (double)c->P * npairs, [&] {
            launch_grid_filter(s, prev, w, h, npairs, c->d_pts, c->P, c->prm.filter_variant, d_mask, d_v, nullptr);
        });
    }
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int check_size(ofarn_ctx *c, int w, int h)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (w < 1 || h < 1) return fail(OFARN_E_INVALID, "empty frame %dx%d", w, h);
    if ((size_t)w * h > (size_t)c->max_w * c->max_h)
        return fail(OFARN_E_SIZE, "frame %dx%d exceeds the context's %dx%d", w, h, c->max_w, c->max_h);
    return OFARN_OK;
}

int ensure_staging(ofarn_ctx *c, size_t frames_bytes, size_t flow_bytes, size_t dm_bytes)
{
    if (frames_bytes > c->st_frames_cap) {
        if (c->st_frames) (void)hipFree(c->st_frames);
        c->st_frames = nullptr; c->st_frames_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->st_frames, frames_bytes));
        c->st_frames_cap = frames_bytes;
    }
    if (flow_bytes > c->st_flow_cap) {
        if (c->st_flow) (void)hipFree(c->st_flow);
        c->st_flow = nullptr; c->st_flow_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->st_flow, flow_bytes));
        c->st_flow_cap = flow_bytes;
    }
    if (dm_bytes > c->st_dm_cap) {
        if (c->st_mask) (void)hipFree(c->st_mask);
        if (c->st_v) (void)hipFree(c->st_v);
        c->st_mask = c->st_v = nullptr; c->st_dm_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->st_mask, dm_bytes));
        HIP_TRY(hipMalloc((void **)&c->st_v, dm_bytes));
        c->st_dm_cap = dm_bytes;
    }
    return OFARN_OK;
}

// Grow-only reservation (sizes in floats; 0 = leave alone).  Growing frees the old buffer first, which waits for
// the device: it happens on the first call of a shape, not per call.
bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

int ws_reserve(ofarn_ctx *c, int wi, size_t tmp, size_t I, size_t R, size_t M, size_t flow, bool capturing)
{
    ofarn_ctx::Workspace &ws = c->ws[wi];
    auto grow = [&](float **p, size_t *cap, size_t need, const char *what) -> int {
        if (need <= *cap) return OFARN_OK;
        if (capturing)
            return fail(OFARN_E_INVALID, "workspace buffer %s would have to grow while the stream is being captured: run the call once "
                        "(or ofarn_reserve) before capturing", what);
        if (*p) { (void)hipFree(*p); c->ws_bytes -= *cap * sizeof(float) + 256; *p = nullptr; *cap = 0; }
        const size_t bytes = need * sizeof(float) + 256;
        if (hipMalloc((void **)p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            *p = nullptr;
            return fail(OFARN_E_NOMEM, "workspace buffer %s of %zu bytes does not fit (max_batch=%d at %dx%d)", what, bytes,
                        c->max_batch, c->max_w, c->max_h);
        }
        *cap = need;
        c->ws_bytes += bytes;
        return OFARN_OK;
    };
    int rc;
    if ((rc = grow(&ws.tmp, &ws.cap_tmp, tmp, "tmp")) || (rc = grow(&ws.I, &ws.cap_I, I, "I")) ||
        (rc = grow(&ws.R, &ws.cap_R, R, "R")) || (rc = grow(&ws.M, &ws.cap_M, M, "M")))
        return rc;
    if (flow > ws.cap_flow) {
        size_t cap = ws.cap_flow;
        if ((rc = grow(&ws.flowA, &cap, flow, "flowA"))) return rc;
        if ((rc = grow(&ws.flowB, &ws.cap_flow, flow, "flowB"))) return rc;
    }
    return OFARN_OK;
}

// Workspace `wi` for max_batch pairs at max_w x max_h: R and the two flow buffers now, the rest on demand.
// Returns 0 on success.
int alloc_workspace(ofarn_ctx *c, int wi)
{
    const size_t px = (size_t)c->max_w * c->max_h;
    const size_t F = (size_t)2 * c->max_batch, Pn = (size_t)c->max_batch;
    if (ws_reserve(c, wi, 0, 0, F * (px * 5 + 4), 0, Pn * px * 2)) return 1;
    if (!c->aux[wi]) {
        if (hipStreamCreateWithFlags(&c->aux[wi], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join[wi], hipEventDisableTiming) != hipSuccess) {
            fail(OFARN_E_HIP, "stream/event creation failed");
            return 1;
        }
    }
    if (!c->ev_fork && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) {
        fail(OFARN_E_HIP, "event creation failed");
        return 1;
    }
    return 0;
}

int begin_call(ofarn_ctx *c, hipStream_t s)
{
    if (c->have_last && c->last_stream != s) HIP_TRY(hipStreamWaitEvent(s, c->ev_done, 0));
    return OFARN_OK;
}

int end_call(ofarn_ctx *c, hipStream_t s)
{
    HIP_TRY(hipEventRecord(c->ev_done, s));
    c->last_stream = s;
    c->have_last = true;
    return OFARN_OK;
}

}  // namespace ofarn_host

extern "C" {
#pragma GCC visibility push(default)

void ofarn_default_params(ofarn_params *p)
{
    if (!p) return;
    p->pyr_scale = 0.5; p->levels = 3; p->winsize = 15; p->iterations = 3;
    p->poly_n = 5; p->poly_sigma = 1.2; p->flags = 0; p->grid_step = 30; p->filter_variant = 0;
}

const char *ofarn_last_error(void) { return g_err.c_str(); }

const char *ofarn_version(void) { return "ofarn 0.3.0 gfx950 (HIP, fp-contract=off)"; }

int ofarn_create(const ofarn_params *params, int device, int max_w, int max_h, int max_batch, ofarn_ctx **out)
{
    if (!out) return fail(OFARN_E_INVALID, "out is NULL");
    *out = nullptr;
    int rc = check_params(params);
    if (rc) return rc;
    if (max_w < 1 || max_h < 1 || max_batch < 1)
        return fail(OFARN_E_INVALID, "max_w, max_h, max_batch must be >= 1 (got %d, %d, %d)", max_w, max_h, max_batch);
    // the kernels address a frame's planes with 32-bit byte offsets (16 bytes per pixel in the widest plane)
    if ((unsigned long long)max_w * (unsigned long long)max_h >= (1ull << 27))
        return fail(OFARN_E_SIZE, "frames of %dx%d exceed the 2^27-pixel limit of the 32-bit plane offsets", max_w, max_h);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(OFARN_E_HIP, "no HIP device visible; libofarn has no CPU path");
    if (device < 0 || device >= ndev) return fail(OFARN_E_INVALID, "device %d out of range [0, %d)", device, ndev);
    OFARN_ON_DEVICE(device);
    ofarn_ctx *c = new ofarn_ctx();
    c->prm = *params;
    c->device = device;
    c->max_w = max_w; c->max_h = max_h; c->max_batch = max_batch;
    {
        const char *e = getenv("OFARN_FORCE_GENERIC");
        c->force_generic = e && e[0] == '1';
    }
    if (const char *e = getenv("OFARN_DIRECT_MIN_FRAMES")) c->direct_min_frames = atoi(e);
    if (const char *e = getenv("OFARN_ROW_LTR")) c->row_small_symm = e[0] == '1' ? 0 : 1;
    if (const char *e = getenv("OFARN_TILE")) c->tile_mode = e[0] != '0';
    if (const char *e = getenv("OFARN_STREAM_ZERO_COPY")) c->stream_zero_copy = e[0] != '0';
    if (const char *e = getenv("OFARN_BOX_ORDER")) c->box_running = e[0] == '1';
    if (!poly_prepare(params->poly_n, params->poly_sigma, c->poly)) {
        delete c;
        return fail(OFARN_E_INVALID, "poly_n out of range");
    }
    auto bail = [&](int code) { ofarn_destroy(c); return code; };
    // The context's own stream (host entry points) gets the HIGHEST stream priority, the internal side streams stay at the default,
    // the pipelined session's copy stream takes the lowest: streams of one priority share a small pool of hardware queues, and two of
    // a context's streams on one queue serialise what the event fork / join was meant to overlap (measured on the pipelined frame
    // loop: 0.58 ms per frame, or 0.69 ... 1.3 ms, depending only on how many streams the process had created before).
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_greatest = 0; }
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) { (void)hipGetLastError(); c->stream = nullptr; }
    if ((!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess)
        return bail(fail(OFARN_E_HIP, "stream/event creation failed"));
    {
        const char *e = getenv("OFARN_SINGLE_STREAM");
        c->dual = !(e && e[0] == '1');
    }
    if (alloc_workspace(c, 0)) return bail(OFARN_E_NOMEM);
    if (params->flags & OFARN_FLAG_FARNEBACK_GAUSSIAN) {
        // FarnebackUpdateFlow_GaussianBlur: kernel[0] = 1, kernel[i] = (float)exp(-i*i/(2 sigma^2)), normalised
        const int m = params->winsize / 2;
        const double sigma = m * 0.3;
        double sum = 1;
        std::vector<float> k(m + 1);
        k[0] = (float)sum;
        for (int i = 1; i <= m; i++) {
            const float t = (float)std::exp(-i * i / (2 * sigma * sigma));
            k[i] = t;
            sum += t * 2;
        }
        sum = 1. / sum;
        for (int i = 0; i <= m; i++) k[i] = (float)(k[i] * sum);
        c->h_gwin = k;
        if (hipMalloc((void **)&c->d_gwin, (m + 1) * sizeof(float)) != hipSuccess ||
            hipMemcpy(c->d_gwin, k.data(), (m + 1) * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(OFARN_E_HIP, "allocating the Gaussian window failed"));
    }
    *out = c;
    return OFARN_OK;
}

void ofarn_destroy(ofarn_ctx *c)
{
    if (!c) return;
    DeviceScope on_device(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_plan(c);
    for (auto &ws : c->ws)
        for (float *p : {ws.tmp, ws.I, ws.R, ws.M, ws.flowA, ws.flowB}) if (p) (void)hipFree(p);
    if (c->st_flow) (void)hipFree(c->st_flow);
    if (c->d_gwin) (void)hipFree(c->d_gwin);
    {
        ofarn_ctx::Stream &st = c->stream_state;
        if (st.copy_stream) { (void)hipStreamSynchronize(st.copy_stream); (void)hipStreamDestroy(st.copy_stream); }
        if (st.h_view) (void)hipHostFree(st.h_view);
        for (int i = 0; i < ofarn_ctx::Stream::kRing; i++) {
            if (st.ring[i]) (void)hipFree(st.ring[i]);
            if (st.ev_computed[i]) (void)hipEventDestroy(st.ev_computed[i]);
            if (st.ev_copied[i]) (void)hipEventDestroy(st.ev_copied[i]);
        }
        for (int i = 0; i < 2; i++) {
            if (st.ev_uploaded[i]) (void)hipEventDestroy(st.ev_uploaded[i]);
            if (st.h_stage[i]) (void)hipHostFree(st.h_stage[i]);
            if (st.h_keep[i]) (void)hipHostFree(st.h_keep[i]);
        }
        if (st.ev_src_uploaded) (void)hipEventDestroy(st.ev_src_uploaded);
    }
    if (c->stream_state.R) (void)hipFree(c->stream_state.R);
    for (uint8_t *p : {c->stream_state.d_frame, c->stream_state.d_bgr, c->stream_state.d_view, c->stream_state.d_lamps}) if (p) (void)hipFree(p);
    for (hipEvent_t e : c->ev_level) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) {
        if (c->aux[i]) { (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamDestroy(c->aux[i]); }
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (uint8_t *p : {c->st_frames, c->st_mask, c->st_v, c->gray[0], c->gray[1]}) if (p) (void)hipFree(p);
    for (uint8_t *p : c->lk.pyr) if (p) (void)hipFree(p);
    for (int16_t *p : c->lk.der) if (p) (void)hipFree(p);
    for (auto &r : c->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : c->prof_free) (void)hipEventDestroy(e);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int ofarn_set_option(ofarn_ctx *c, const char *name, int value)
{
    if (!c || !name) return fail(OFARN_E_INVALID, "ctx or name is NULL");
    const std::string n(name);
    if (n == "tile") c->tile_mode = value < 0 ? -1 : (value != 0);
    else if (n == "force_generic") c->force_generic = value != 0;
    else if (n == "row_ltr") c->row_small_symm = value ? 0 : 1;
    else if (n == "direct_min_frames") c->direct_min_frames = value;
    else if (n == "single_stream") c->dual = value == 0;
    else if (n == "stream_zero_copy") c->stream_zero_copy = value != 0;
    else if (n == "stream_overlap") c->stream_overlap = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "push_blocks") c->push_blocks = value < 0 ? 0 : value;
    else if (n == "debug_fail_wave") c->debug_fail_wave = value;
    else if (n == "prof_dual") c->prof_dual = value != 0;
    else if (n == "box_order") c->box_running = value != 0;
    else return fail(OFARN_E_INVALID, "unknown option '%s'", name);
    return OFARN_OK;
}

int ofarn_reserve(ofarn_ctx *c, int w, int h, int n_pairs, int pairs_mode)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (n_pairs < 1) return fail(OFARN_E_INVALID, "n_pairs must be >= 1");
    if (pairs_mode != OFARN_PAIRS_INDEPENDENT && pairs_mode != OFARN_PAIRS_CONSECUTIVE)
        return fail(OFARN_E_INVALID, "pairs_mode must be 0 or 1");
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    const int np = n_pairs < c->max_batch ? n_pairs : c->max_batch;          // a wave never holds more
    const int nframes = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? np + 1 : 2 * np;
    const bool gauss = (c->prm.flags & OFARN_FLAG_FARNEBACK_GAUSSIAN) != 0;
    const bool fused = !c->force_generic && !(c->box_running && !gauss) && c->prm.iterations >= 1 &&
                       (gauss ? flow_iter_gauss_supported(c->prm.winsize) : flow_iter_supported(c->prm.winsize));
    const size_t M_need = fused ? 0 : (size_t)np * w * h * ((c->box_running && !gauss) ? 15 : 5);
    const int nws = (c->dual && n_pairs > c->max_batch) ? 2 : 1;
    for (int wi = 0; wi < nws; wi++) {
        if (wi == 1 && alloc_workspace(c, 1)) return OFARN_E_NOMEM;
        // frames aligned for the direct level kernels (what hipMalloc / torch give) and not: reserve for the larger of the two
        size_t tmp = 0, I = 0;
        for (uintptr_t a : {(uintptr_t)256, (uintptr_t)1}) {
            WavePlan wp;
            plan_wave(c, reinterpret_cast<const uint8_t *>(a), nframes, w, h, wp);
            tmp = std::max(tmp, wp.tmp_need);
            I = std::max(I, wp.I_need);
        }
        if ((rc = ws_reserve(c, wi, tmp, I, 0, M_need, 0))) return rc;
    }
    return OFARN_OK;
}

double ofarn_last_device_ms(const ofarn_ctx *c) { return c ? c->last_ms : 0; }
uint64_t ofarn_workspace_bytes(const ofarn_ctx *c) { return c ? c->ws_bytes : 0; }

int ofarn_profile_enable(ofarn_ctx *c, int on)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    c->prof_on = on != 0;
    return OFARN_OK;
}

int ofarn_profile_read(ofarn_ctx *c, int cap, int *stage, int *level, int *launches, double *ms, double *units)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    OFARN_ON_DEVICE(c->device);
    for (auto &r : c->prof_pending) {
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        if (r.level >= 0 && r.level < 32) {
            auto &a = c->prof_acc[r.stage][r.level];
            a.launches++; a.ms += t; a.units += r.units;
        }
        c->prof_free.push_back(r.a);
        c->prof_free.push_back(r.b);
    }
    c->prof_pending.clear();
    int n = 0;
    for (int st = 0; st < OFARN_STAGE_COUNT; st++)
        for (int lv = 0; lv < 32; lv++) {
            auto &a = c->prof_acc[st][lv];
            if (!a.launches) continue;
            if (n < cap) {
                if (stage) stage[n] = st;
                if (level) level[n] = lv;
                if (launches) launches[n] = a.launches;
                if (ms) ms[n] = a.ms;
                if (units) units[n] = a.units;
            }
            n++;
            a = ofarn_ctx::ProfAcc();
        }
    return n;
}

int ofarn_level_plan(const ofarn_params *p, int w, int h, int cap, int *lw, int *lh, int *ksize, double *sigma)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (w < 1 || h < 1) return fail(OFARN_E_INVALID, "empty frame");
    const int nlev = crop_levels(w, h, p->pyr_scale, p->levels);
    for (int k = 0; k <= nlev && k < cap; k++) {
        int a, b, ks;
        double sg;
        level_geom(w, h, p->pyr_scale, k, a, b, sg, ks);
        if (lw) lw[k] = a;
        if (lh) lh[k] = b;
        if (ksize) ksize[k] = ks;
        if (sigma) sigma[k] = sg;
    }
    return nlev + 1;
}

int ofarn_grid_points(int w, int h, int step, float *h_pts)
{
    if (w < 1 || h < 1 || step < 1) return fail(OFARN_E_INVALID, "bad grid arguments");
    std::vector<int> xs, ys;
    axis_points(w, step, &xs);
    axis_points(h, step, &ys);
    if (h_pts) {
        size_t i = 0;
        for (int x : xs)
            for (int y : ys) { h_pts[i++] = (float)x; h_pts[i++] = (float)y; }
    }
    return (int)(xs.size() * ys.size());
}

#pragma GCC visibility pop
}  // extern "C"

namespace ofarn_host {

// gray staging for BGR input: 2 * max_batch frames per workspace
int ensure_gray(ofarn_ctx *c, int wi)
{
    if (c->gray[wi]) return OFARN_OK;
    const size_t bytes = (size_t)2 * c->max_batch * c->max_w * c->max_h + 256;
    if (hipMalloc((void **)&c->gray[wi], bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OFARN_E_NOMEM, "gray staging of %zu bytes does not fit", bytes);
    }
    c->ws_bytes += bytes;
    return OFARN_OK;
}


int calc_batch_device_impl(ofarn_ctx *c, const uint8_t *d_frames, bool bgr, int n_frames, int w, int h, int pairs_mode,
                           float *d_flow, uint8_t *d_mask, uint8_t *d_v, void *hip_stream)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!d_frames) return fail(OFARN_E_INVALID, "frames is NULL");
    if (pairs_mode != OFARN_PAIRS_INDEPENDENT && pairs_mode != OFARN_PAIRS_CONSECUTIVE)
        return fail(OFARN_E_INVALID, "pairs_mode must be 0 or 1");
    const int n_pairs = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? n_frames - 1 : n_frames / 2;
    if (n_pairs < 0 || (pairs_mode == OFARN_PAIRS_INDEPENDENT && (n_frames & 1)))
        return fail(OFARN_E_INVALID, "n_frames=%d does not form whole pairs in mode %d", n_frames, pairs_mode);
    if (n_pairs == 0) return OFARN_OK;
    const bool use_init = (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) != 0;
    if (use_init && !d_flow) return fail(OFARN_E_INVALID, "OPTFLOW_USE_INITIAL_FLOW needs the flow buffer (it holds the initial flow)");
    OFARN_ON_DEVICE(c->device);
    hipStream_t s = pick_stream(c, hip_stream);
    // Recorded into a HIP graph?  Then nothing below may allocate, free or copy synchronously: refuse BEFORE the first such
    // call (it would invalidate the caller's capture) whatever a warm-up call or ofarn_reserve would have prepared.
    const bool capturing = stream_is_capturing(s);
    if (capturing && (c->plan_w != w || c->plan_h != h || (bgr && !c->gray[0])))
        return fail(OFARN_E_INVALID, "the stream is being captured and this context has not seen %dx%d%s frames yet: run the call once, "
                    "or ofarn_reserve, before capturing", w, h, bgr ? " BGR" : "");
    if ((rc = make_plan(c, w, h))) return rc;
    if ((rc = begin_call(c, s))) return rc;
    const size_t fsz = (size_t)w * h;
    const int fstep = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? 1 : 2;
    const int nwaves = (n_pairs + c->max_batch - 1) / c->max_batch;
    if (bgr && (rc = ensure_gray(c, 0))) return rc;
    // More than one wave: alternate them over two internal streams (each with its own workspace), forked
    // from and joined back into the caller's stream with events.  Per-kernel profiling keeps one stream.  (During capture
    // only if the second workspace already exists.)
    const bool dual = c->dual && nwaves > 1 && (!c->prof_on || c->prof_dual) && (!capturing || (c->ws[1].R && (!bgr || c->gray[1]))) &&
                      alloc_workspace(c, 1) == 0 && (!bgr || ensure_gray(c, 1) == 0);
    if (dual) {
        HIP_TRY(hipEventRecord(c->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(c->aux[0], c->ev_fork, 0));
        HIP_TRY(hipStreamWaitEvent(c->aux[1], c->ev_fork, 0));
    }
    int wi = 0;
    for (int p0 = 0; p0 < n_pairs; p0 += c->max_batch, wi ^= 1) {
        const int np = n_pairs - p0 < c->max_batch ? n_pairs - p0 : c->max_batch;
        hipStream_t ws_stream = dual ? c->aux[wi] : s;
        const uint8_t *wave_frames = d_frames + (size_t)p0 * fstep * fsz * (bgr ? 3 : 1);
        if (bgr) {
            // frame front end (DenseOF.py:481,510): BGR frames of this wave -> gray, then the usual path
            const int nf = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? np + 1 : 2 * np;
            uint8_t *g = c->gray[dual ? wi : 0];
            timed(c, ws_stream, OFARN_STAGE_BGR2GRAY, 0, (double)nf * fsz, [&] {
                launch_bgr2gray(ws_stream, wave_frames, g, (size_t)nf * fsz, kGrayB, kGrayG, kGrayR, kGrayShift);
            });
            wave_frames = g;
        }
        float *wave_flow = d_flow ? d_flow + (size_t)p0 * fsz * 2 : nullptr;
        rc = run_wave(c, ws_stream, wave_frames, np, pairs_mode, w, h, wave_flow,
                      d_mask ? d_mask + (size_t)p0 * c->P : nullptr, d_v ? d_v + (size_t)p0 * c->P : nullptr,
                      dual ? wi : 0, use_init ? wave_flow : nullptr);
        if (rc) break;      // waves already enqueued keep running: the join and the call's event below must still happen
    }
    // Join the two internal streams back into the caller's and record the call's event on EVERY path, the failing one
    // included: an earlier wave may still be running on an internal stream with the shared workspace, and the next call on
    // this context (or a stream capture the caller has open) must be ordered behind it.
    int jrc = OFARN_OK;
    if (dual)
        for (int i = 0; i < 2; i++)
            if (hipEventRecord(c->ev_join[i], c->aux[i]) != hipSuccess || hipStreamWaitEvent(s, c->ev_join[i], 0) != hipSuccess) {
                (void)hipGetLastError();
                if (!rc && !jrc) jrc = fail(OFARN_E_HIP, "joining the internal streams failed");
            }
    const int erc = end_call(c, s);
    return rc ? rc : (jrc ? jrc : erc);
}


}  // namespace ofarn_host

extern "C" {
#pragma GCC visibility push(default)

int ofarn_calc_batch_device(ofarn_ctx *c, const uint8_t *d_frames, int n_frames, int w, int h, int pairs_mode,
                            float *d_flow, uint8_t *d_mask, uint8_t *d_v, void *hip_stream)
{
    return calc_batch_device_impl(c, d_frames, false, n_frames, w, h, pairs_mode, d_flow, d_mask, d_v, hip_stream);
}

int ofarn_calc_batch_device_bgr(ofarn_ctx *c, const uint8_t *d_bgr, int n_frames, int w, int h, int pairs_mode,
                                float *d_flow, uint8_t *d_mask, uint8_t *d_v, void *hip_stream)
{
    return calc_batch_device_impl(c, d_bgr, true, n_frames, w, h, pairs_mode, d_flow, d_mask, d_v, hip_stream);
}

int ofarn_calc_batch(ofarn_ctx *c, const uint8_t *h_frames, int n_frames, int w, int h, int pairs_mode,
                     float *h_flow, uint8_t *h_mask, uint8_t *h_v)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_frames) return fail(OFARN_E_INVALID, "frames is NULL");
    if (pairs_mode != OFARN_PAIRS_INDEPENDENT && pairs_mode != OFARN_PAIRS_CONSECUTIVE)
        return fail(OFARN_E_INVALID, "pairs_mode must be 0 or 1");
    const int n_pairs = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? n_frames - 1 : n_frames / 2;
    if (n_pairs < 0 || (pairs_mode == OFARN_PAIRS_INDEPENDENT && (n_frames & 1)))
        return fail(OFARN_E_INVALID, "n_frames=%d does not form whole pairs in mode %d", n_frames, pairs_mode);
    if ((h_mask == nullptr) != (h_v == nullptr)) return fail(OFARN_E_INVALID, "danger mask and v must be given together");
    if (n_pairs == 0) return OFARN_OK;
    const bool use_init = (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) != 0;
    if (use_init && !h_flow) return fail(OFARN_E_INVALID, "OPTFLOW_USE_INITIAL_FLOW needs the flow buffer (it holds the initial flow)");
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    const size_t fsz = (size_t)w * h;
    const int fstep = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? 1 : 2;
    const int wave = c->max_batch;
    const size_t wave_frames = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? wave + 1 : 2 * (size_t)wave;
    if ((rc = ensure_staging(c, wave_frames * fsz, (size_t)wave * fsz * 2 * sizeof(float),
                             h_mask ? (size_t)wave * (c->P > 0 ? c->P : 1) : 0)))
        return rc;
    double ms_total = 0;
    if ((rc = begin_call(c, c->stream))) return rc;
    for (int p0 = 0; p0 < n_pairs; p0 += wave) {
        const int np = n_pairs - p0 < wave ? n_pairs - p0 : wave;
        const int nf = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? np + 1 : 2 * np;
        HIP_TRY(hipMemcpyAsync(c->st_frames, h_frames + (size_t)p0 * fstep * fsz, (size_t)nf * fsz,
                               hipMemcpyHostToDevice, c->stream));
        if (use_init)
            HIP_TRY(hipMemcpyAsync(c->st_flow, h_flow + (size_t)p0 * fsz * 2, (size_t)np * fsz * 2 * sizeof(float),
                                   hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        rc = run_wave(c, c->stream, c->st_frames, np, pairs_mode, w, h, c->st_flow, h_mask ? c->st_mask : nullptr,
                      h_mask ? c->st_v : nullptr, 0, use_init ? c->st_flow : nullptr);
        if (rc) { (void)end_call(c, c->stream); return rc; }
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        if (h_flow)
            HIP_TRY(hipMemcpyAsync(h_flow + (size_t)p0 * fsz * 2, c->st_flow, (size_t)np * fsz * 2 * sizeof(float),
                                   hipMemcpyDeviceToHost, c->stream));
        if (h_mask && c->P > 0) {
            HIP_TRY(hipMemcpyAsync(h_mask + (size_t)p0 * c->P, c->st_mask, (size_t)np * c->P, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(h_v + (size_t)p0 * c->P, c->st_v, (size_t)np * c->P, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        ms_total += ms;
    }
    c->last_ms = ms_total;
    return end_call(c, c->stream);
}

int ofarn_calc(ofarn_ctx *c, const uint8_t *h_prev, const uint8_t *h_next, int w, int h, int stride, float *h_flow)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_prev || !h_next || !h_flow) return fail(OFARN_E_INVALID, "prev, next and flow must not be NULL");
    if (stride < w) return fail(OFARN_E_INVALID, "stride %d < width %d", stride, w);
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    const size_t fsz = (size_t)w * h;
    if ((rc = ensure_staging(c, 2 * fsz, fsz * 2 * sizeof(float), 0))) return rc;
    if ((rc = begin_call(c, c->stream))) return rc;
    if (stride == w) {      // dense rows: plain copies (a 2-D copy of pageable memory is staged row by row)
        HIP_TRY(hipMemcpyAsync(c->st_frames, h_prev, fsz, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->st_frames + fsz, h_next, fsz, hipMemcpyHostToDevice, c->stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(c->st_frames, w, h_prev, stride, w, h, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpy2DAsync(c->st_frames + fsz, w, h_next, stride, w, h, hipMemcpyHostToDevice, c->stream));
    }
    const bool use_init = (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) != 0;
    if (use_init)   // cv2: `flow` is an in/out argument holding the initial flow
        HIP_TRY(hipMemcpyAsync(c->st_flow, h_flow, fsz * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    // a page-locked flow buffer (ofarn_host_alloc / hipHostMalloc) is written by the last iteration kernel itself, as in
    // ofarn_stream_next: no copy behind the kernels ("stream_zero_copy" = 0 switches it off)
    float *d_out = c->st_flow;
    bool direct = false;
    if (!use_init && c->stream_zero_copy) {
        if (void *dp = mapped_host_range(h_flow, fsz * 2 * sizeof(float))) {
            d_out = static_cast<float *>(dp);
            direct = true;
        }
    }
    if ((rc = run_wave(c, c->stream, c->st_frames, 1, OFARN_PAIRS_INDEPENDENT, w, h, d_out, nullptr, nullptr, 0,
                       use_init ? c->st_flow : nullptr))) {
        (void)end_call(c, c->stream);
        return rc;
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    if (!direct) HIP_TRY(hipMemcpyAsync(h_flow, c->st_flow, fsz * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    return end_call(c, c->stream);
}

int ofarn_grid_filter_device(ofarn_ctx *c, const float *d_flow, int n, int w, int h, uint8_t *d_mask, uint8_t *d_v,
                             int32_t *d_iflow, void *hip_stream)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!d_flow || !d_mask || !d_v) return fail(OFARN_E_INVALID, "flow, mask and v must not be NULL");
    if (n < 0) return fail(OFARN_E_INVALID, "n < 0");
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    if (c->P == 0) return OFARN_OK;
    hipStream_t s = pick_stream(c, hip_stream);
    launch_grid_filter(s, d_flow, w, h, n, c->d_pts, c->P, c->prm.filter_variant, d_mask, d_v, d_iflow);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_grid_filter(ofarn_ctx *c, const float *h_flow, int n, int w, int h, uint8_t *h_mask, uint8_t *h_v,
                      int32_t *h_iflow)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_flow || !h_mask || !h_v) return fail(OFARN_E_INVALID, "flow, mask and v must not be NULL");
    if (n < 0) return fail(OFARN_E_INVALID, "n < 0");
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    if (c->P == 0) return OFARN_OK;
    const size_t fsz = (size_t)w * h * 2 * sizeof(float);
    if ((rc = ensure_staging(c, 0, fsz, (size_t)c->P))) return rc;
    DevTmp iflow_tmp;
    if (h_iflow && (rc = iflow_tmp.alloc((size_t)c->P * 2 * sizeof(int32_t)))) return rc;
    int32_t *d_if = iflow_tmp.as<int32_t>();
    if ((rc = begin_call(c, c->stream))) return rc;
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipMemcpyAsync(c->st_flow, h_flow + (size_t)i * w * h * 2, fsz, hipMemcpyHostToDevice, c->stream));
        launch_grid_filter(c->stream, c->st_flow, w, h, 1, c->d_pts, c->P, c->prm.filter_variant, c->st_mask, c->st_v, d_if);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_mask + (size_t)i * c->P, c->st_mask, c->P, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_v + (size_t)i * c->P, c->st_v, c->P, hipMemcpyDeviceToHost, c->stream));
        if (h_iflow)
            HIP_TRY(hipMemcpyAsync(h_iflow + (size_t)i * c->P * 2, d_if, (size_t)c->P * 2 * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return end_call(c, c->stream);
}


// (sparse LK entry points: ofarn_api_lk.hip)

#pragma GCC visibility pop
}  // extern "C"
