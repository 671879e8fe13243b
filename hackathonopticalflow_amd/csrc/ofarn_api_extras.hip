// ofarn_api_extras.hip -- C-ABI entry points either side of the dense-flow call: BGR -> gray, the dense
// visualisers, and the single-stage entry points the parity tests use.  Shared internals: ofarn_host.h.
#include "ofarn_host.h"

using namespace ofarn;
using namespace ofarn_host;

namespace ofarn_host {

// np.mgrid[step/2:size:step] (DenseOF.py:44): count and float start
int arrow_axis(int size, int step, double *start)
{
    *start = step / 2.0;
    const int n = (int)std::ceil((size - *start) / (step * 1.0));
    return n < 0 ? 0 : n;
}

// cv2.circle(img, center, radius, color, thickness=-1) with LINE_8 and shift 0 runs drawing.cpp's Circle(): a midpoint loop that
// fills the spans [cx - dx, cx + dx] on rows cy +- dy and [cx - dy, cx + dy] on rows cy +- dx.  ext[d] = the widest span half-width
// any step gives row d (radius 6: 6 5 5 5 4 3 0).
static void circle_extents(int radius, uint8_t *ext)
{
    for (int i = 0; i <= radius; i++) ext[i] = 0;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > ext[dy]) ext[dy] = (uint8_t)dx;
        if (dy > ext[dx]) ext[dx] = (uint8_t)dy;
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

int lamp_grid(const ofarn_ctx *c, int w, int h, int radius, ofarn::LampGrid *g, int *P)
{
    if (radius < 0 || radius > kMaxLampRadius) return fail(OFARN_E_INVALID, "lamp radius %d outside 0..%d", radius, kMaxLampRadius);
    if (c->prm.grid_step <= 2 * radius)
        return fail(OFARN_E_UNSUPPORTED, "discs of radius %d on a grid of step %d would overlap (the later disc then takes the colour "
                    "under its centre, pathfinder_viewer.py:220-222): not built", radius, c->prm.grid_step);
    std::vector<int> xs, ys;
    axis_points(w, c->prm.grid_step, &xs);
    axis_points(h, c->prm.grid_step, &ys);
    g->x0 = xs.empty() ? 0 : xs[0];
    g->y0 = ys.empty() ? 0 : ys[0];
    g->step = c->prm.grid_step;
    g->nx = (int)xs.size();
    g->ny = (int)ys.size();
    g->radius = radius;
    circle_extents(radius, g->ext);
    *P = g->nx * g->ny;
    return OFARN_OK;
}

}  // namespace ofarn_host

extern "C" {
#pragma GCC visibility push(default)

int ofarn_bgr2gray_device(ofarn_ctx *c, const uint8_t *d_bgr, int n, int w, int h, uint8_t *d_gray, void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!d_bgr || !d_gray) return fail(OFARN_E_INVALID, "bgr and gray must not be NULL");
    if (n < 0 || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad size n=%d %dx%d", n, w, h);
    OFARN_ON_DEVICE(c->device);
    hipStream_t s = pick_stream(c, hip_stream);
    launch_bgr2gray(s, d_bgr, d_gray, (size_t)n * w * h, kGrayB, kGrayG, kGrayR, kGrayShift);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_bgr2gray(ofarn_ctx *c, const uint8_t *h_bgr, int n, int w, int h, int stride, uint8_t *h_gray)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_bgr || !h_gray) return fail(OFARN_E_INVALID, "bgr and gray must not be NULL");
    if (n < 0 || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad size n=%d %dx%d", n, w, h);
    if (stride < 3 * w) return fail(OFARN_E_INVALID, "stride %d < 3 * width %d", stride, w);
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    const size_t npx = (size_t)w * h;
    DevTmp in, out;
    int rc;
    if ((rc = in.alloc(npx * 3)) || (rc = out.alloc(npx))) return rc;
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipMemcpy2DAsync(in.p, (size_t)w * 3, h_bgr + (size_t)i * stride * h, stride, (size_t)w * 3, h,
                                 hipMemcpyHostToDevice, c->stream));
        launch_bgr2gray(c->stream, in.as<uint8_t>(), out.as<uint8_t>(), npx, kGrayB, kGrayG, kGrayR, kGrayShift);
        HIP_TRY(hipMemcpyAsync(h_gray + (size_t)i * npx, out.p, npx, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return OFARN_OK;
}

int ofarn_flow_hsv_device(ofarn_ctx *c, const float *d_flow, int n, int w, int h, uint8_t *d_hsv, uint8_t *d_bgr, void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!d_flow || (!d_hsv && !d_bgr)) return fail(OFARN_E_INVALID, "flow and at least one of hsv, bgr must not be NULL");
    if (n < 0 || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad size n=%d %dx%d", n, w, h);
    OFARN_ON_DEVICE(c->device);
    hipStream_t s = pick_stream(c, hip_stream);
    launch_flow_hsv(s, d_flow, (size_t)n * w * h, d_hsv, d_bgr);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_flow_hsv(ofarn_ctx *c, const float *h_flow, int n, int w, int h, uint8_t *h_hsv, uint8_t *h_bgr)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_flow || (!h_hsv && !h_bgr)) return fail(OFARN_E_INVALID, "flow and at least one of hsv, bgr must not be NULL");
    if (n < 0 || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad size n=%d %dx%d", n, w, h);
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    const size_t npx = (size_t)w * h;
    DevTmp in, o1, o2;
    int rc;
    if ((rc = in.alloc(npx * 8)) || (rc = o1.alloc(npx * 3)) || (rc = o2.alloc(npx * 3))) return rc;
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipMemcpyAsync(in.p, h_flow + (size_t)i * npx * 2, npx * 8, hipMemcpyHostToDevice, c->stream));
        launch_flow_hsv(c->stream, in.as<float>(), npx, h_hsv ? o1.as<uint8_t>() : nullptr, h_bgr ? o2.as<uint8_t>() : nullptr);
        if (h_hsv) HIP_TRY(hipMemcpyAsync(h_hsv + (size_t)i * npx * 3, o1.p, npx * 3, hipMemcpyDeviceToHost, c->stream));
        if (h_bgr) HIP_TRY(hipMemcpyAsync(h_bgr + (size_t)i * npx * 3, o2.p, npx * 3, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return OFARN_OK;
}

int ofarn_hsv2bgr(ofarn_ctx *c, const uint8_t *h_hsv, size_t npx, uint8_t *h_bgr)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_hsv || !h_bgr) return fail(OFARN_E_INVALID, "hsv and bgr must not be NULL");
    if (npx == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    DevTmp in, out;
    int rc;
    if ((rc = in.alloc(npx * 3)) || (rc = out.alloc(npx * 3))) return rc;
    HIP_TRY(hipMemcpyAsync(in.p, h_hsv, npx * 3, hipMemcpyHostToDevice, c->stream));
    launch_hsv2bgr(c->stream, in.as<uint8_t>(), npx, out.as<uint8_t>());
    HIP_TRY(hipMemcpyAsync(h_bgr, out.p, npx * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OFARN_OK;
}

int ofarn_draw_flow_device(ofarn_ctx *c, const float *d_flow, int n, int w, int h, int step, const uint8_t *d_base, uint8_t *d_out,
                           void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!d_flow || !d_out) return fail(OFARN_E_INVALID, "flow and out must not be NULL");
    if (n < 0 || w < 1 || h < 1 || step < 1) return fail(OFARN_E_INVALID, "bad arguments n=%d %dx%d step=%d", n, w, h, step);
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    double st;
    const int nx = arrow_axis(w, step, &st), ny = arrow_axis(h, step, &st);
    const size_t bytes = (size_t)n * w * h * 3;
    hipStream_t s = pick_stream(c, hip_stream);
    if (!d_base) HIP_TRY(hipMemsetAsync(d_out, 0, bytes, s));
    else if (d_base != d_out) HIP_TRY(hipMemcpyAsync(d_out, d_base, bytes, hipMemcpyDeviceToDevice, s));
    launch_draw_flow(s, d_flow, w, h, n, nx, ny, st, (double)step, d_out);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_draw_flow(ofarn_ctx *c, const float *h_flow, int n, int w, int h, int step, const uint8_t *h_base, uint8_t *h_out)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_flow || !h_out) return fail(OFARN_E_INVALID, "flow and out must not be NULL");
    if (n < 0 || w < 1 || h < 1 || step < 1) return fail(OFARN_E_INVALID, "bad arguments n=%d %dx%d step=%d", n, w, h, step);
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    double st;
    const int nx = arrow_axis(w, step, &st), ny = arrow_axis(h, step, &st);
    const size_t npx = (size_t)w * h, img = npx * 3;
    DevTmp in, out;
    int rc;
    if ((rc = in.alloc(npx * 8)) || (rc = out.alloc(img))) return rc;
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipMemcpyAsync(in.p, h_flow + (size_t)i * npx * 2, npx * 8, hipMemcpyHostToDevice, c->stream));
        if (h_base) HIP_TRY(hipMemcpyAsync(out.p, h_base + (size_t)i * img, img, hipMemcpyHostToDevice, c->stream));
        else HIP_TRY(hipMemsetAsync(out.p, 0, img, c->stream));
        launch_draw_flow(c->stream, in.as<float>(), w, h, 1, nx, ny, st, (double)step, out.as<uint8_t>());
        HIP_TRY(hipMemcpyAsync(h_out + (size_t)i * img, out.p, img, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return OFARN_OK;
}

int ofarn_add_u8_device(ofarn_ctx *c, const uint8_t *d_a, const uint8_t *d_b, size_t n, uint8_t *d_out, void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (n > 0 && (!d_a || !d_b || !d_out)) return fail(OFARN_E_INVALID, "a, b and out must not be NULL");
    OFARN_ON_DEVICE(c->device);
    launch_add_u8(pick_stream(c, hip_stream), d_a, d_b, d_out, n);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_add_u8(ofarn_ctx *c, const uint8_t *h_a, const uint8_t *h_b, size_t n, uint8_t *h_out)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (n == 0) return OFARN_OK;
    if (!h_a || !h_b || !h_out) return fail(OFARN_E_INVALID, "a, b and out must not be NULL");
    OFARN_ON_DEVICE(c->device);
    DevTmp a, b;
    int rc;
    if ((rc = a.alloc(n)) || (rc = b.alloc(n))) return rc;
    HIP_TRY(hipMemcpyAsync(a.p, h_a, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b.p, h_b, n, hipMemcpyHostToDevice, c->stream));
    launch_add_u8(c->stream, a.as<uint8_t>(), b.as<uint8_t>(), a.as<uint8_t>(), n);
    HIP_TRY(hipMemcpyAsync(h_out, a.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OFARN_OK;
}

int ofarn_draw_lamps_device(ofarn_ctx *c, const uint8_t *d_mask, const uint8_t *d_v, int n, int w, int h, int radius,
                            const uint8_t *d_base, uint8_t *d_out, void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!d_out) return fail(OFARN_E_INVALID, "out is NULL");
    if (n < 0 || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad arguments n=%d %dx%d", n, w, h);
    LampGrid g;
    int P = 0;
    int rc = lamp_grid(c, w, h, radius, &g, &P);
    if (rc) return rc;
    if (P > 0 && (!d_mask || !d_v)) return fail(OFARN_E_INVALID, "mask and v must not be NULL");
    OFARN_ON_DEVICE(c->device);
    launch_draw_lamps(pick_stream(c, hip_stream), d_mask, d_v, P, d_base, d_out, w, h, n, g);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_draw_lamps(ofarn_ctx *c, const uint8_t *h_mask, const uint8_t *h_v, int n, int w, int h, int radius,
                     const uint8_t *h_base, uint8_t *h_out)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_out) return fail(OFARN_E_INVALID, "out is NULL");
    if (n < 0 || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad arguments n=%d %dx%d", n, w, h);
    LampGrid g;
    int P = 0;
    int rc = lamp_grid(c, w, h, radius, &g, &P);
    if (rc) return rc;
    if (P > 0 && (!h_mask || !h_v)) return fail(OFARN_E_INVALID, "mask and v must not be NULL");
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    const size_t img = (size_t)w * h * 3;
    DevTmp dm, base, out;
    if ((rc = dm.alloc((size_t)2 * P + 16)) || (rc = out.alloc(img)) || (h_base && (rc = base.alloc(img)))) return rc;
    for (int i = 0; i < n; i++) {
        if (P > 0) {
            HIP_TRY(hipMemcpyAsync(dm.p, h_mask + (size_t)i * P, P, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(dm.as<uint8_t>() + P, h_v + (size_t)i * P, P, hipMemcpyHostToDevice, c->stream));
        }
        if (h_base) HIP_TRY(hipMemcpyAsync(base.p, h_base + (size_t)i * img, img, hipMemcpyHostToDevice, c->stream));
        launch_draw_lamps(c->stream, dm.as<uint8_t>(), dm.as<uint8_t>() + P, P, h_base ? base.as<uint8_t>() : nullptr, out.as<uint8_t>(),
                          w, h, 1, g);
        HIP_TRY(hipMemcpyAsync(h_out + (size_t)i * img, out.p, img, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return OFARN_OK;
}

int ofarn_flow_arrow_count(int w, int h, int step, int *nx, int *ny)
{
    if (w < 1 || h < 1 || step < 1) return fail(OFARN_E_INVALID, "bad arrow grid arguments");
    double st;
    const int ax = arrow_axis(w, step, &st), ay = arrow_axis(h, step, &st);
    if (nx) *nx = ax;
    if (ny) *ny = ay;
    return ax * ay;
}

int ofarn_flow_arrows_device(ofarn_ctx *c, const float *d_flow, int n, int w, int h, int step, int32_t *d_lines, void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!d_flow || !d_lines) return fail(OFARN_E_INVALID, "flow and lines must not be NULL");
    if (n < 0 || w < 1 || h < 1 || step < 1) return fail(OFARN_E_INVALID, "bad arguments n=%d %dx%d step=%d", n, w, h, step);
    OFARN_ON_DEVICE(c->device);
    double st;
    const int nx = arrow_axis(w, step, &st), ny = arrow_axis(h, step, &st);
    hipStream_t s = pick_stream(c, hip_stream);
    launch_flow_arrows(s, d_flow, w, h, n, nx, ny, st, (double)step, d_lines);
    HIP_TRY(hipGetLastError());
    return OFARN_OK;
}

int ofarn_flow_arrows(ofarn_ctx *c, const float *h_flow, int n, int w, int h, int step, int32_t *h_lines)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_flow || !h_lines) return fail(OFARN_E_INVALID, "flow and lines must not be NULL");
    if (n < 0 || w < 1 || h < 1 || step < 1) return fail(OFARN_E_INVALID, "bad arguments n=%d %dx%d step=%d", n, w, h, step);
    if (n == 0) return OFARN_OK;
    OFARN_ON_DEVICE(c->device);
    double st;
    const int nx = arrow_axis(w, step, &st), ny = arrow_axis(h, step, &st);
    const size_t npx = (size_t)w * h, K = (size_t)nx * ny;
    if (K == 0) return OFARN_OK;
    DevTmp in, out;
    int rc;
    if ((rc = in.alloc(npx * 8)) || (rc = out.alloc(K * 4 * sizeof(int32_t)))) return rc;
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipMemcpyAsync(in.p, h_flow + (size_t)i * npx * 2, npx * 8, hipMemcpyHostToDevice, c->stream));
        launch_flow_arrows(c->stream, in.as<float>(), w, h, 1, nx, ny, st, (double)step, out.as<int32_t>());
        HIP_TRY(hipMemcpyAsync(h_lines + (size_t)i * K * 4, out.p, K * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return OFARN_OK;
}

int ofarn_stage_resize_area(ofarn_ctx *c, const float *h_flow, int sw, int sh, int dw, int dh, float mul, float *h_out)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_flow || !h_out) return fail(OFARN_E_INVALID, "NULL argument");
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return fail(OFARN_E_INVALID, "bad sizes");
    OFARN_ON_DEVICE(c->device);
    const size_t saved = c->plan_allocs.size();
    AreaTabHost t;
    int rc = build_area_tab(c, sw, sh, dw, dh, t);
    DevTmp in, out;
    if (!rc) rc = in.alloc((size_t)sw * sh * 8);
    if (!rc) rc = out.alloc((size_t)dw * dh * 8);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(in.p, h_flow, (size_t)sw * sh * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            launch_resize_area(c->stream, in.as<float>(), sw, sh, out.as<float>(), dw, dh, 1, t, mul);
            e = hipMemcpyAsync(h_out, out.p, (size_t)dw * dh * 8, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(OFARN_E_HIP, "resize_area stage failed: %s", hipGetErrorString(e));
    }
    while (c->plan_allocs.size() > saved) { (void)hipFree(c->plan_allocs.back()); c->plan_allocs.pop_back(); }
    return rc;
}

int ofarn_stage_pyrdown(ofarn_ctx *c, const uint8_t *h_img, int w, int h, uint8_t *h_out)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_img || !h_out || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad argument");
    OFARN_ON_DEVICE(c->device);
    const size_t npx = (size_t)w * h, nout = (size_t)((w + 1) / 2) * ((h + 1) / 2);
    DevTmp in, out;
    int rc;
    if ((rc = in.alloc(npx)) || (rc = out.alloc(nout))) return rc;
    HIP_TRY(hipMemcpyAsync(in.p, h_img, npx, hipMemcpyHostToDevice, c->stream));
    launch_pyrdown_u8(c->stream, in.as<uint8_t>(), w, h, out.as<uint8_t>(), 1);
    HIP_TRY(hipMemcpyAsync(h_out, out.p, nout, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OFARN_OK;
}

int ofarn_stage_scharr(ofarn_ctx *c, const uint8_t *h_img, int w, int h, int16_t *h_out)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (!h_img || !h_out || w < 1 || h < 1) return fail(OFARN_E_INVALID, "bad argument");
    OFARN_ON_DEVICE(c->device);
    const size_t npx = (size_t)w * h;
    DevTmp in, out;
    int rc;
    if ((rc = in.alloc(npx)) || (rc = out.alloc(npx * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(in.p, h_img, npx, hipMemcpyHostToDevice, c->stream));
    launch_scharr(c->stream, in.as<uint8_t>(), w, h, out.as<int16_t>(), 0, 1, 1);
    HIP_TRY(hipMemcpyAsync(h_out, out.p, npx * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OFARN_OK;
}

// ------------------------------------------------------------------ single-stage entry points
// Each uses the context workspace for one image; inputs/outputs are host arrays.

int ofarn_stage_level_image(ofarn_ctx *c, const uint8_t *h_img, int w, int h, int k, float *h_out)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_img || !h_out) return fail(OFARN_E_INVALID, "NULL argument");
    OFARN_ON_DEVICE(c->device);
    if ((rc = make_plan(c, w, h))) return rc;
    if (k < 0 || k >= (int)c->lv.size()) return fail(OFARN_E_INVALID, "level %d out of range", k);
    const Level &L = c->lv[k];
    const size_t fsz = (size_t)w * h;
    if ((rc = ensure_staging(c, fsz, 0, 0))) return rc;
    if ((rc = ws_reserve(c, 0, (size_t)h * L.w * 2, (size_t)L.w * L.h, 0, 0, 0))) return rc;
    if ((rc = begin_call(c, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c->st_frames, h_img, fsz, hipMemcpyHostToDevice, c->stream));
    const bool lds_ok = (size_t)(w + 2 * (L.ksize / 2)) * 4 * 33 / 32 + 4 * (size_t)L.ksize + 64 <= 60 * 1024;
    if (!c->force_generic && level_direct_supported(c->st_frames, w, h, L.w, L.h, L.ksize))
        launch_level_direct(c->stream, c->st_frames, fsz, w, h, 1, L.h_kern.data(), L.ksize, c->ws[0].I, L.w, L.h, c->row_small_symm);
    else {
        if (!c->force_generic && level_hdirect_supported(c->st_frames, w, L.w, L.ksize))
            launch_level_hdirect(c->stream, c->st_frames, fsz, w, h, 1, L.h_kern.data(), L.ksize, c->ws[0].tmp, L.w);
        else if (!c->force_generic && lds_ok)
            launch_level_hpass_lds(c->stream, c->st_frames, fsz, w, h, 1, L.d_kern, L.ksize, L.d_xofs, L.w, c->ws[0].tmp, c->row_small_symm);
        else
            launch_level_hpass(c->stream, c->st_frames, fsz, w, h, 1, L.d_kern, L.ksize, L.d_xofs, L.w, c->ws[0].tmp, c->row_small_symm);
        launch_level_vpass(c->stream, c->ws[0].tmp, h, L.w, L.h, 1, L.d_kern, L.ksize, L.d_xa, L.d_yofs, L.d_ya, c->ws[0].I);
    }
    HIP_TRY(hipMemcpyAsync(h_out, c->ws[0].I, (size_t)L.w * L.h * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return end_call(c, c->stream);
}

// host <-> device layout helpers of the single-stage entry points (test-only paths)
static void r_to_device_layout(const float *h_il, size_t npx, std::vector<float> &dev)   // [npx][5] -> 4+1
{
    dev.assign(r_frame_stride(npx), 0.f);
    for (size_t o = 0; o < npx; o++) {
        for (int ch = 0; ch < 4; ch++) dev[o * 4 + ch] = h_il[o * 5 + ch];
        dev[4 * npx + o] = h_il[o * 5 + 4];
    }
}
static void r_from_device_layout(const std::vector<float> &dev, size_t npx, float *h_il)
{
    for (size_t o = 0; o < npx; o++) {
        for (int ch = 0; ch < 4; ch++) h_il[o * 5 + ch] = dev[o * 4 + ch];
        h_il[o * 5 + 4] = dev[4 * npx + o];
    }
}

int ofarn_stage_polyexp(ofarn_ctx *c, const float *h_img, int w, int h, float *h_R)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_img || !h_R) return fail(OFARN_E_INVALID, "NULL argument");
    OFARN_ON_DEVICE(c->device);
    const size_t npx = (size_t)w * h;
    if ((rc = ws_reserve(c, 0, 0, npx, 0, 0, 0))) return rc;
    if ((rc = begin_call(c, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c->ws[0].I, h_img, npx * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (!c->force_generic && polyexp_march_supported(c->prm.poly_n)) {
        const float none[3] = {0, 0, 0};
        launch_polyexp_march(c->stream, c->ws[0].I, npx, 0, c->ws[0].R, w, h, 1, c->poly, none);
    } else
        launch_polyexp(c->stream, c->ws[0].I, c->ws[0].R, w, h, 1, c->poly);
    std::vector<float> dev(r_frame_stride(npx));
    HIP_TRY(hipMemcpyAsync(dev.data(), c->ws[0].R, dev.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    r_from_device_layout(dev, npx, h_R);
    return end_call(c, c->stream);
}

int ofarn_stage_update_matrices(ofarn_ctx *c, const float *h_R0, const float *h_R1, const float *h_flow, int w, int h,
                                float *h_M)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_R0 || !h_R1 || !h_flow || !h_M) return fail(OFARN_E_INVALID, "NULL argument");
    OFARN_ON_DEVICE(c->device);
    const size_t npx = (size_t)w * h;
    std::vector<float> d0, d1;
    r_to_device_layout(h_R0, npx, d0);
    r_to_device_layout(h_R1, npx, d1);
    if ((rc = ws_reserve(c, 0, 0, 0, 0, npx * 5, 0))) return rc;
    if ((rc = begin_call(c, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c->ws[0].R, d0.data(), d0.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ws[0].R + r_frame_stride(npx), d1.data(), d1.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ws[0].flowA, h_flow, npx * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    launch_update_matrices(c->stream, c->ws[0].R, 1, c->ws[0].flowA, c->ws[0].M, w, h, 1);
    std::vector<float> mp(npx * 5);
    HIP_TRY(hipMemcpyAsync(mp.data(), c->ws[0].M, npx * 5 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t o = 0; o < npx; o++)
        for (int ch = 0; ch < 5; ch++) h_M[o * 5 + ch] = mp[ch * npx + o];     // planar -> interleaved
    return end_call(c, c->stream);
}

int ofarn_stage_blur_solve(ofarn_ctx *c, const float *h_M, int w, int h, float *h_flow)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (!h_M || !h_flow) return fail(OFARN_E_INVALID, "NULL argument");
    OFARN_ON_DEVICE(c->device);
    const size_t npx = (size_t)w * h;
    std::vector<float> mp(npx * 5);
    for (size_t o = 0; o < npx; o++)
        for (int ch = 0; ch < 5; ch++) mp[ch * npx + o] = h_M[o * 5 + ch];     // interleaved -> planar
    const bool gauss = (c->prm.flags & OFARN_FLAG_FARNEBACK_GAUSSIAN) != 0;
    const bool running = c->box_running && !gauss;          // "box_order" = 1: the literal order; its double column sums sit in front of M
    if ((rc = ws_reserve(c, 0, 0, 0, 0, npx * (running ? 15 : 5), 0))) return rc;
    if ((rc = begin_call(c, c->stream))) return rc;
    float *Mk = running ? c->ws[0].M + npx * 10 : c->ws[0].M;
    HIP_TRY(hipMemcpyAsync(Mk, mp.data(), npx * 5 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (gauss)
        launch_gauss_solve(c->stream, Mk, c->ws[0].flowA, w, h, 1, c->prm.winsize, c->d_gwin);
    else if (running)
        launch_blur_solve_running(c->stream, Mk, reinterpret_cast<double *>(c->ws[0].M), c->ws[0].flowA, w, h, 1, c->prm.winsize);
    else
        launch_blur_solve(c->stream, Mk, c->ws[0].flowA, w, h, 1, c->prm.winsize);
    HIP_TRY(hipMemcpyAsync(h_flow, c->ws[0].flowA, npx * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return end_call(c, c->stream);
}

int ofarn_stage_flow_upsample(ofarn_ctx *c, const float *h_flow, int sw, int sh, int dw, int dh, float *h_out)
{
    int rc = check_size(c, dw, dh);
    if (rc) return rc;
    if ((rc = check_size(c, sw, sh))) return rc;
    if (!h_flow || !h_out) return fail(OFARN_E_INVALID, "NULL argument");
    OFARN_ON_DEVICE(c->device);
    std::vector<int> xo, yo;
    std::vector<float> xa, ya;
    resize_tables(sw, dw, xo, xa);
    resize_tables(sh, dh, yo, ya);
    DevTmp t_xo, t_xa, t_yo, t_ya;
    if ((rc = t_xo.alloc(dw * sizeof(int))) || (rc = t_xa.alloc(dw * sizeof(float))) || (rc = t_yo.alloc(dh * sizeof(int))) ||
        (rc = t_ya.alloc(dh * sizeof(float))))
        return rc;
    int *d_xo = t_xo.as<int>(), *d_yo = t_yo.as<int>();
    float *d_xa = t_xa.as<float>(), *d_ya = t_ya.as<float>();
    HIP_TRY(hipMemcpy(d_xo, xo.data(), dw * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_xa, xa.data(), dw * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_yo, yo.data(), dh * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ya, ya.data(), dh * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = begin_call(c, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c->ws[0].flowA, h_flow, (size_t)sw * sh * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    launch_flow_upsample(c->stream, c->ws[0].flowA, sw, sh, c->ws[0].flowB, dw, dh, 1, d_xo, d_xa, d_yo, d_ya,
                         (float)(1. / c->prm.pyr_scale));
    HIP_TRY(hipMemcpyAsync(h_out, c->ws[0].flowB, (size_t)dw * dh * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return end_call(c, c->stream);
}

#pragma GCC visibility pop
}  // extern "C"
