// ofarn_api_stream.hip -- streaming session of the C-ABI (include/ofarn.h, ofarn_stream_*): the reference's frame loop
//     gray = cvtColor(img); flow = calculate_optical_flow(prev_gray, gray); prev_gray = gray      DenseOF.py:510, 519-525
// hands over ONE new frame per turn.  A context keeps the previous frame's polynomial expansions of every pyramid level on the
// device (two slots per level, the newest frame goes into the slot the older one does not occupy), so a turn uploads one frame,
// runs stages A + B once and the iterations of the pair; nothing of the previous frame is re-uploaded or recomputed.
// Results are those of ofarn_calc on (previous frame, new frame), bit for bit.
#include "ofarn_host.h"

#include <map>
#include <mutex>

using namespace ofarn;
using namespace ofarn_host;

namespace ofarn_host {

// Blocks handed out by ofarn_host_alloc: base address -> size.  A pointer into one of them is checked against the block's extent
// here; page-locked memory from elsewhere (hipHostMalloc, hipHostRegister, a torch pin_memory() tensor) is looked up in the runtime.
static std::mutex g_pins_mu;
static std::map<uintptr_t, size_t> g_pins;

// "The GPU may write [p, p + bytes) through the returned pointer" -- or nullptr.  Round 3 asked only whether p was page-locked: a
// pointer into the tail of a smaller page-locked block passed and became an out-of-bounds device write.
void *mapped_host_range(const void *p, size_t bytes)
{
    if (!p) return nullptr;
    const uintptr_t a = (uintptr_t)p;
    bool known = false, covered = false;
    {
        std::lock_guard<std::mutex> lk(g_pins_mu);
        auto it = g_pins.upper_bound(a);
        if (it != g_pins.begin()) {
            --it;
            if (a < it->first + it->second) { known = true; covered = bytes <= it->second && a - it->first <= it->second - bytes; }
        }
    }
    if (known && !covered) return nullptr;
    hipPointerAttribute_t at;
    void *dp = nullptr;
    if (!(hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost &&
          hipHostGetDevicePointer(&dp, const_cast<void *>(p), 0) == hipSuccess && dp)) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (known) return dp;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dp) != hipSuccess || !base) {
        (void)hipGetLastError();
        return nullptr;                        // extent unknown: the copy path is always right
    }
    const uintptr_t b = (uintptr_t)base, d = (uintptr_t)dp;
    if (d < b || bytes > size || d - b > size - bytes) return nullptr;
    return dp;
}

}  // namespace ofarn_host

namespace {

// Two R slots per level for the plan of (w, h); resets the session when the frame size changes.
int stream_prepare(ofarn_ctx *c, int w, int h)
{
    ofarn_ctx::Stream &st = c->stream_state;
    if (st.w == w && st.h == h && st.R && st.off.size() == c->lv.size()) return OFARN_OK;
    st.have = false;
    st.cur = 0;
    st.w = w; st.h = h;
    st.off.assign(c->lv.size(), 0);
    size_t need = 0;
    for (size_t k = 0; k < c->lv.size(); k++) {
        st.off[k] = need;
        need += 2 * r_frame_stride((size_t)c->lv[k].w * c->lv[k].h);
    }
    if (need > st.cap) {
        if (st.R) { (void)hipFree(st.R); c->ws_bytes -= st.cap * sizeof(float) + 256; st.R = nullptr; st.cap = 0; }
        const size_t bytes = need * sizeof(float) + 256;
        if (hipMalloc((void **)&st.R, bytes) != hipSuccess) {
            (void)hipGetLastError();
            st.R = nullptr;
            st.w = st.h = 0;
            return fail(OFARN_E_NOMEM, "streaming state of %zu bytes does not fit", bytes);
        }
        st.cap = need;
        c->ws_bytes += bytes;
    }
    return OFARN_OK;
}

int grow_u8(ofarn_ctx *c, uint8_t **p, size_t *cap, size_t need, const char *what)
{
    if (need <= *cap) return OFARN_OK;
    if (*p) { (void)hipFree(*p); c->ws_bytes -= *cap + 256; *p = nullptr; *cap = 0; }
    if (hipMalloc((void **)p, need + 256) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        return fail(OFARN_E_NOMEM, "%s of %zu bytes does not fit", what, need);
    }
    *cap = need;
    c->ws_bytes += need + 256;
    return OFARN_OK;
}

// One turn on stream s with the new gray frame already on the device.  Returns OFARN_OK (flow written) or
// OFARN_STREAM_PRIMED (first frame of a session: nothing to pair it with).
int stream_turn(ofarn_ctx *c, hipStream_t s, const uint8_t *d_gray, int w, int h, float *d_flow, uint8_t *d_mask, uint8_t *d_v,
                bool pair = false)
{
    ofarn_ctx::Stream &st = c->stream_state;
    if (pair) st.have = false;                   // pair turn: d_gray holds both frames, the session starts over with them
    const bool had = st.have || pair;
    const bool use_init = (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) != 0;
    int rc = run_wave(c, s, d_gray, 1, OFARN_PAIRS_CONSECUTIVE, w, h, d_flow, had ? d_mask : nullptr, had ? d_v : nullptr, 0,
                      (use_init && had) ? d_flow : nullptr, &st, pair);
    if (rc) {
        st.have = false;                         // a half-written slot must not be paired with anything
        if (c->aux[0]) (void)hipStreamSynchronize(c->aux[0]);   // stages A + B may have been left running beside the caller's stream
        return rc;
    }
    st.cur = pair ? 1 : (had ? st.cur ^ 1 : 0);
    st.have = true;
    st.turns++;
    st.view_flow_valid = had && d_flow == c->st_flow && d_flow != nullptr;
    st.view_danger_valid = st.view_bgr_valid = false;      // ofarn_stream_next_view sets them behind its own turn
    return had ? OFARN_OK : OFARN_STREAM_PRIMED;
}

int stream_check(ofarn_ctx *c, int w, int h, hipStream_t s = nullptr, bool device_entry = false)
{
    int rc = check_size(c, w, h);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));        // the entry point's OFARN_ON_DEVICE scope restores the caller's device
    const ofarn_ctx::Stream &st = c->stream_state;
    if (device_entry && stream_is_capturing(s) && (c->plan_w != w || c->plan_h != h || st.w != w || st.h != h || !st.R))
        return fail(OFARN_E_INVALID, "the stream is being captured and this context's streaming session has not seen %dx%d frames yet: "
                    "run one turn before capturing", w, h);
    if ((rc = make_plan(c, w, h))) return rc;
    return stream_prepare(c, w, h);
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int ofarn_stream_reset(ofarn_ctx *c)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    c->stream_state.have = false;
    c->stream_state.cur = 0;
    return OFARN_OK;
}

int ofarn_stream_primed(const ofarn_ctx *c, int w, int h)
{
    if (!c) return 0;
    const ofarn_ctx::Stream &st = c->stream_state;
    return st.have && st.w == w && st.h == h;
}

int ofarn_stream_next_device(ofarn_ctx *c, const uint8_t *d_gray, int w, int h, float *d_flow, uint8_t *d_mask, uint8_t *d_v,
                             void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    OFARN_ON_DEVICE(c->device);
    int rc = stream_check(c, w, h, pick_stream(c, hip_stream), true);
    if (rc) return rc;
    if (!d_gray) return fail(OFARN_E_INVALID, "frame is NULL");
    if ((d_mask == nullptr) != (d_v == nullptr)) return fail(OFARN_E_INVALID, "danger mask and v must be given together");
    if (c->stream_state.have && !d_flow && !d_mask) return fail(OFARN_E_INVALID, "flow and danger maps are all NULL: nothing to compute");
    if ((c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) && c->stream_state.have && !d_flow)
        return fail(OFARN_E_INVALID, "OPTFLOW_USE_INITIAL_FLOW needs the flow buffer (it holds the initial flow)");
    hipStream_t s = pick_stream(c, hip_stream);
    if ((rc = begin_call(c, s))) return rc;
    const int turn = stream_turn(c, s, d_gray, w, h, d_flow, d_mask, d_v);
    rc = end_call(c, s);
    return turn < 0 ? turn : (rc ? rc : turn);
}

int ofarn_stream_next_device_bgr(ofarn_ctx *c, const uint8_t *d_bgr, int w, int h, float *d_flow, uint8_t *d_mask, uint8_t *d_v,
                                 void *hip_stream)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    OFARN_ON_DEVICE(c->device);
    int rc = stream_check(c, w, h, pick_stream(c, hip_stream), true);
    if (rc) return rc;
    if (!d_bgr) return fail(OFARN_E_INVALID, "frame is NULL");
    if ((d_mask == nullptr) != (d_v == nullptr)) return fail(OFARN_E_INVALID, "danger mask and v must be given together");
    ofarn_ctx::Stream &st = c->stream_state;
    const size_t fsz = (size_t)w * h;
    if ((rc = grow_u8(c, &st.d_frame, &st.frame_cap, fsz, "streaming frame buffer"))) return rc;
    hipStream_t s = pick_stream(c, hip_stream);
    if ((rc = begin_call(c, s))) return rc;
    timed(c, s, OFARN_STAGE_BGR2GRAY, 0, (double)fsz, [&] { launch_bgr2gray(s, d_bgr, st.d_frame, fsz, kGrayB, kGrayG, kGrayR, kGrayShift); });
    const int turn = stream_turn(c, s, st.d_frame, w, h, d_flow, d_mask, d_v);
    rc = end_call(c, s);
    return turn < 0 ? turn : (rc ? rc : turn);
}

// Host frame in, host flow out; synchronous.  `bgr` != 0: h_frame is packed BGR (3 bytes per pixel, `stride` bytes per row).
static int stream_next_host(ofarn_ctx *c, const uint8_t *h_frame, int bgr, int w, int h, int stride, float *h_flow,
                            uint8_t *h_mask, uint8_t *h_v)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    OFARN_ON_DEVICE(c->device);
    int rc = stream_check(c, w, h);
    if (rc) return rc;
    const int bpp = bgr ? 3 : 1;
    if (!h_frame) return fail(OFARN_E_INVALID, "frame is NULL");
    if (stride < w * bpp) return fail(OFARN_E_INVALID, "stride %d < row bytes %d", stride, w * bpp);
    if ((h_mask == nullptr) != (h_v == nullptr)) return fail(OFARN_E_INVALID, "danger mask and v must be given together");
    ofarn_ctx::Stream &st = c->stream_state;
    const bool had = st.have;
    if (had && !h_flow) return fail(OFARN_E_INVALID, "flow is NULL");
    const size_t fsz = (size_t)w * h;
    if ((rc = grow_u8(c, &st.d_frame, &st.frame_cap, fsz, "streaming frame buffer"))) return rc;
    if (bgr && (rc = grow_u8(c, &st.d_bgr, &st.bgr_cap, fsz * 3, "streaming BGR buffer"))) return rc;
    if ((rc = ensure_staging(c, 0, fsz * 2 * sizeof(float), h_mask ? (size_t)(c->P > 0 ? c->P : 1) : 0))) return rc;
    hipStream_t s = c->stream;
    if ((rc = begin_call(c, s))) return rc;
    uint8_t *dst = bgr ? st.d_bgr : st.d_frame;
    if (stride == w * bpp) HIP_TRY(hipMemcpyAsync(dst, h_frame, fsz * bpp, hipMemcpyHostToDevice, s));
    else HIP_TRY(hipMemcpy2DAsync(dst, (size_t)w * bpp, h_frame, stride, (size_t)w * bpp, h, hipMemcpyHostToDevice, s));
    const bool use_init = (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) != 0;
    if (use_init && had) HIP_TRY(hipMemcpyAsync(c->st_flow, h_flow, fsz * 2 * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(c->ev0, s));
    if (bgr)
        timed(c, s, OFARN_STAGE_BGR2GRAY, 0, (double)fsz, [&] { launch_bgr2gray(s, st.d_bgr, st.d_frame, fsz, kGrayB, kGrayG, kGrayR, kGrayShift); });
    // a pinned (ofarn_host_alloc / hipHostMalloc) flow buffer is written by the last iteration kernel itself: no copy behind it
    float *d_out = c->st_flow;
    bool direct = false;
    if (had && !use_init && c->stream_zero_copy) {
        if (void *dp = mapped_host_range(h_flow, fsz * 2 * sizeof(float))) {
            d_out = static_cast<float *>(dp);
            direct = true;
        }
    }
    const int turn = stream_turn(c, s, st.d_frame, w, h, d_out, h_mask ? c->st_mask : nullptr, h_mask ? c->st_v : nullptr);
    if (turn < 0) { (void)end_call(c, s); return turn; }
    HIP_TRY(hipEventRecord(c->ev1, s));
    if (turn == OFARN_OK) {
        if (!direct) HIP_TRY(hipMemcpyAsync(h_flow, c->st_flow, fsz * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
        if (h_mask && c->P > 0) {
            HIP_TRY(hipMemcpyAsync(h_mask, c->st_mask, c->P, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(h_v, c->st_v, c->P, hipMemcpyDeviceToHost, s));
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    rc = end_call(c, s);
    return rc ? rc : turn;
}

int ofarn_stream_next(ofarn_ctx *c, const uint8_t *h_gray, int w, int h, int stride, float *h_flow)
{
    return stream_next_host(c, h_gray, 0, w, h, stride, h_flow, nullptr, nullptr);
}

int ofarn_stream_next_bgr(ofarn_ctx *c, const uint8_t *h_bgr, int w, int h, int stride, float *h_flow)
{
    return stream_next_host(c, h_bgr, 1, w, h, stride, h_flow, nullptr, nullptr);
}

int ofarn_stream_next_danger(ofarn_ctx *c, const uint8_t *h_gray, int w, int h, int stride, float *h_flow, uint8_t *h_mask,
                             uint8_t *h_v)
{
    return stream_next_host(c, h_gray, 0, w, h, stride, h_flow, h_mask, h_v);
}

// ofarn_calc for a caller that hands over consecutive pairs (DenseOF.py:519-525: flow = calculate_optical_flow(prev_gray, gray);
// prev_gray = gray), with ofarn_calc's contract: the result is a function of (prev, next) alone, for EVERY input.  The context
// keeps, page-locked on the host, a byte copy of the frame it was last given as `next` (the copy doubles as the upload's staging
// buffer, which a pageable source needs anyway).  When `prev` equals that copy -- all w x h bytes compared, no fingerprint -- the
// frame already on the device with its level images and polynomial expansions is reused and only `next` is uploaded and
// expanded; otherwise both frames are.  The comparison runs on the host WHILE the device already computes the turn it would allow
// (the optimistic turn is simply redone from both frames if the comparison fails), so exactness costs no latency in the loop.
// With OPTFLOW_USE_INITIAL_FLOW (h_flow is in/out) the comparison comes first.  *reused (optional): 1 if the frame was reused.
int ofarn_calc_reuse(ofarn_ctx *c, const uint8_t *h_prev, const uint8_t *h_next, int w, int h, int stride_prev, int stride_next,
                     float *h_flow, int *reused)
{
    if (reused) *reused = 0;
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    OFARN_ON_DEVICE(c->device);
    int rc = stream_check(c, w, h);
    if (rc) return rc;
    if (!h_prev || !h_next || !h_flow) return fail(OFARN_E_INVALID, "prev, next and flow must not be NULL");
    if (stride_prev < w || stride_next < w) return fail(OFARN_E_INVALID, "stride %d / %d < width %d", stride_prev, stride_next, w);
    ofarn_ctx::Stream &st = c->stream_state;
    const size_t fsz = (size_t)w * h, flow_bytes = fsz * 2 * sizeof(float);
    if ((rc = grow_u8(c, &st.d_frame, &st.frame_cap, 2 * fsz, "streaming frame buffer"))) return rc;     // room for a pair (miss path)
    if ((rc = ensure_staging(c, 0, flow_bytes, 0))) return rc;
    hipStream_t s = c->stream;
    if (fsz > st.keep_cap) {
        for (int i = 0; i < 2; i++)
            if (st.h_keep[i]) { (void)hipHostFree(st.h_keep[i]); st.h_keep[i] = nullptr; }
        st.keep_cap = 0;
        st.keep_valid = false;
        for (int i = 0; i < 2; i++)
            if (hipHostMalloc((void **)&st.h_keep[i], fsz, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                st.h_keep[i] = nullptr;
                return fail(OFARN_E_NOMEM, "page-locked frame copies of %zu bytes could not be allocated", fsz);
            }
        st.keep_cap = fsz;
    }
    if ((rc = begin_call(c, s))) return rc;
    const bool use_init = (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) != 0;
    // h_keep[keep_cur] is the frame in the session's current slot only if no other entry point has moved the session since
    bool cand = st.have && st.keep_valid && st.keep_turn == st.turns;
    const int kn = cand ? st.keep_cur ^ 1 : 0;             // where `next` is staged; the other one holds (or will hold) `prev`
    auto stage = [&](uint8_t *dst, const uint8_t *src, int stride) {
        if (stride == w) memcpy(dst, src, fsz);
        else for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, src + (size_t)y * stride, (size_t)w);
    };
    auto same_as_kept = [&]() -> bool {
        const uint8_t *kept = st.h_keep[kn ^ 1];
        if (stride_prev == w) return memcmp(kept, h_prev, fsz) == 0;
        for (int y = 0; y < h; y++)
            if (memcmp(kept + (size_t)y * w, h_prev + (size_t)y * stride_prev, (size_t)w) != 0) return false;
        return true;
    };
    // A cheap look before the optimistic launch (a hint only -- the full comparison below decides): 17 rows spread over the frame.
    // A caller who passes unrelated pairs differs here already and goes straight to the both-frames path instead of paying for a
    // turn that is thrown away (1.7 instead of 1.0 ms per call).
    auto sample_same = [&]() -> bool {
        const uint8_t *kept = st.h_keep[kn ^ 1];
        for (int k = 0; k <= 16; k++) {
            const int y = (int)((long long)(h - 1) * k / 16);
            if (memcmp(kept + (size_t)y * w, h_prev + (size_t)y * stride_prev, (size_t)w) != 0) return false;
        }
        return true;
    };
    auto give_up = [&](int code) { st.have = false; st.keep_valid = false; (void)end_call(c, s); return code; };
    float *d_out = c->st_flow;
    bool direct = false;
    if (!use_init && c->stream_zero_copy) {
        if (void *dp = mapped_host_range(h_flow, flow_bytes)) { d_out = static_cast<float *>(dp); direct = true; }
    }
    // A dense `next` is uploaded from where it lies (from pageable memory the call returns once the runtime has staged it) and its
    // byte copy is made afterwards, beside the device's work; a strided one is staged into the copy first and uploaded from there.
    const bool next_dense = stride_next == w;
    bool next_kept = false;
    auto keep_next = [&]() { if (!next_kept) { stage(st.h_keep[kn], h_next, stride_next); next_kept = true; } };
    // one turn of the session with `next` -- or, with both = true, a pair turn from (prev, next): the session starts over and both
    // frames go through the level builds and expansions in ONE set of launches; everything enqueued, nothing waited for
    auto enqueue_pair = [&](bool both) -> int {
        if (both) HIP_TRY(hipMemcpyAsync(st.d_frame, st.h_keep[kn ^ 1], fsz, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(st.d_frame + (both ? fsz : 0), next_kept ? st.h_keep[kn] : h_next, fsz, hipMemcpyHostToDevice, s));
        if (use_init) HIP_TRY(hipMemcpyAsync(c->st_flow, h_flow, flow_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(c->ev0, s));
        const int turn = stream_turn(c, s, st.d_frame, w, h, d_out, nullptr, nullptr, both);
        if (turn < 0) return turn;
        if (turn != OFARN_OK) return fail(OFARN_E_HIP, "internal: the session held no frame to pair with");
        HIP_TRY(hipEventRecord(c->ev1, s));
        if (!direct) HIP_TRY(hipMemcpyAsync(h_flow, c->st_flow, flow_bytes, hipMemcpyDeviceToHost, s));
        return OFARN_OK;
    };
    if (!next_dense) keep_next();
    bool reuse = false;
    if (cand && use_init) cand = reuse = same_as_kept();    // in/out flow: nothing may be overwritten before we know
    else if (cand && !sample_same()) cand = false;
    if (cand) {
        if ((rc = enqueue_pair(false)) < 0) return give_up(rc);
        keep_next();                                        // both beside the device's work
        if (!use_init) reuse = same_as_kept();
    }
    if (!reuse) {
        // `prev` is not what the device holds (or nothing is held): a new session from both frames.  An optimistic turn that may
        // be running is ordered in front of this on the stream and its output is overwritten.
        st.have = false;
        stage(st.h_keep[kn ^ 1], h_prev, stride_prev);
        if ((rc = enqueue_pair(true)) < 0) return give_up(rc);
        keep_next();
        st.reuse_misses++;
    } else st.reuse_hits++;
    if (hipStreamSynchronize(s) != hipSuccess) return give_up(fail(OFARN_E_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError())));
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms; else (void)hipGetLastError();
    st.keep_cur = kn;
    st.keep_valid = true;
    st.keep_turn = st.turns;
    if (reused) *reused = reuse ? 1 : 0;
    return end_call(c, s);
}

int ofarn_calc_reuse_info(const ofarn_ctx *c, unsigned long long *hits, unsigned long long *misses)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    if (hits) *hits = c->stream_state.reuse_hits;
    if (misses) *misses = c->stream_state.reuse_misses;
    return OFARN_OK;
}

// Test hook for the device scope of the entry points (ofarn_host.h, DeviceScope): fake_current >= 0 makes every entry point believe
// the calling thread's current device was that ordinal (no hipSetDevice back to it is issued); -1 restores normal operation.
// Returns through the pointers, for the calling thread: scopes entered so far, and the ordinal the last scope restored (-1: none).
int ofarn_debug_device_scope(int fake_current, int *scopes, int *last_restored)
{
    g_fake_current_device = fake_current;
    if (scopes) *scopes = t_device_scopes;
    if (last_restored) *last_restored = t_last_restored_device;
    t_last_restored_device = -1;
    return OFARN_OK;
}

// 1 if a flow buffer at [p, p + bytes) would be written by the GPU in place (page-locked over its whole extent), 0 if the copy path
// is taken.  For tests: the decision can be checked without letting a kernel write anywhere.
int ofarn_debug_mapped_host_range(const void *p, size_t bytes) { return mapped_host_range(p, bytes) != nullptr; }

// The frame loop's per-frame OUTPUTS without the flow field crossing PCIe.  What the reference does with `flow` each turn is draw it:
// draw_flow samples it on a step-14 grid and draws arrows (DenseOF.py:40-49, :574), draw_hsv paints the rainbow (DenseOF.py:109-124,
// :578), and the danger points come from the grid filter (pathfinder_viewer.py:159-176, 204-217).  Those results are KBs (the arrow
// end points, mask + V) or a third of the flow's size (the rainbow, 3 B instead of 8 B per pixel); the 16.6 MB flow field itself
// stays in HBM (st_flow) -- a synchronous 1080p turn then costs its kernels plus microseconds of transfer instead of + 0.33 ms.
int ofarn_stream_next_view(ofarn_ctx *c, const uint8_t *h_frame, int bgr, int w, int h, int stride, uint8_t *h_mask, uint8_t *h_v,
                           int arrow_step, int32_t *h_lines, uint8_t *h_rainbow)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    OFARN_ON_DEVICE(c->device);
    int rc = stream_check(c, w, h);
    if (rc) return rc;
    const int bpp = bgr ? 3 : 1;
    if (!h_frame) return fail(OFARN_E_INVALID, "frame is NULL");
    if (stride < w * bpp) return fail(OFARN_E_INVALID, "stride %d < row bytes %d", stride, w * bpp);
    if ((h_mask == nullptr) != (h_v == nullptr)) return fail(OFARN_E_INVALID, "danger mask and v must be given together");
    if (h_lines && arrow_step < 1) return fail(OFARN_E_INVALID, "arrow_step must be >= 1, got %d", arrow_step);
    if (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW)
        return fail(OFARN_E_UNSUPPORTED, "ofarn_stream_next_view keeps the flow on the device and has no initial-flow input: use ofarn_stream_next");
    ofarn_ctx::Stream &st = c->stream_state;
    st.view_danger_valid = st.view_bgr_valid = false;      // the buffers below may move
    const size_t fsz = (size_t)w * h;
    double astart = 0;
    const int nx = h_lines ? arrow_axis(w, arrow_step, &astart) : 0, ny = h_lines ? arrow_axis(h, arrow_step, &astart) : 0;
    const size_t K = (size_t)nx * ny, lines_bytes = K * 4 * sizeof(int32_t), rb_bytes = h_rainbow ? fsz * 3 : 0;
    const size_t P = (h_mask && c->P > 0) ? (size_t)c->P : 0;
    // device view buffer: [lines | mask | v | pad to 256 | rainbow]: the small results leave in ONE transfer into a page-locked landing
    // zone (three pageable copies of a few KB each cost ~15 us apiece in call overhead) and are copied on from there by the host
    const size_t small = lines_bytes + 2 * P, small_pad = (small + 255) & ~(size_t)255;
    if ((rc = grow_u8(c, &st.d_frame, &st.frame_cap, fsz, "streaming frame buffer"))) return rc;
    if (bgr && (rc = grow_u8(c, &st.d_bgr, &st.bgr_cap, fsz * 3, "streaming BGR buffer"))) return rc;
    if ((rc = grow_u8(c, &st.d_view, &st.view_cap, small_pad + rb_bytes + 16, "view buffer"))) return rc;
    if ((rc = ensure_staging(c, 0, fsz * 2 * sizeof(float), 0))) return rc;
    if (small > st.h_view_cap) {
        if (st.h_view) { (void)hipHostFree(st.h_view); st.h_view = nullptr; st.h_view_cap = 0; }
        if (hipHostMalloc((void **)&st.h_view, small_pad, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            st.h_view = nullptr;
            return fail(OFARN_E_NOMEM, "page-locked view buffer of %zu bytes could not be allocated", small_pad);
        }
        st.h_view_cap = small_pad;
    }
    int32_t *d_lines = reinterpret_cast<int32_t *>(st.d_view);
    uint8_t *d_mask = st.d_view + lines_bytes, *d_v = d_mask + P, *d_rb = st.d_view + small_pad;
    hipStream_t s = c->stream;
    if ((rc = begin_call(c, s))) return rc;
    uint8_t *dst = bgr ? st.d_bgr : st.d_frame;
    if (stride == w * bpp) HIP_TRY(hipMemcpyAsync(dst, h_frame, fsz * bpp, hipMemcpyHostToDevice, s));
    else HIP_TRY(hipMemcpy2DAsync(dst, (size_t)w * bpp, h_frame, stride, (size_t)w * bpp, h, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(c->ev0, s));
    if (bgr)
        timed(c, s, OFARN_STAGE_BGR2GRAY, 0, (double)fsz, [&] { launch_bgr2gray(s, st.d_bgr, st.d_frame, fsz, kGrayB, kGrayG, kGrayR, kGrayShift); });
    const int turn = stream_turn(c, s, st.d_frame, w, h, c->st_flow, P ? d_mask : nullptr, P ? d_v : nullptr);
    if (turn < 0) { (void)end_call(c, s); return turn; }
    if (turn == OFARN_OK) {
        if (h_lines && K > 0) launch_flow_arrows(s, c->st_flow, w, h, 1, nx, ny, astart, (double)arrow_step, d_lines);
        if (h_rainbow) launch_flow_hsv(s, c->st_flow, fsz, nullptr, d_rb);
        HIP_TRY(hipGetLastError());
    }
    if (turn == OFARN_OK) {
        st.view_danger_valid = P > 0;
        st.view_mask_off = lines_bytes;
        st.view_P = (int)P;
        st.view_bgr_valid = bgr != 0;
    }
    HIP_TRY(hipEventRecord(c->ev1, s));
    if (turn == OFARN_OK) {
        if (small) HIP_TRY(hipMemcpyAsync(st.h_view, st.d_view, small, hipMemcpyDeviceToHost, s));
        if (h_rainbow) HIP_TRY(hipMemcpyAsync(h_rainbow, d_rb, rb_bytes, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (turn == OFARN_OK) {
        if (h_lines && K > 0) memcpy(h_lines, st.h_view, lines_bytes);
        if (P) { memcpy(h_mask, st.h_view + lines_bytes, P); memcpy(h_v, st.h_view + lines_bytes + P, P); }
    }
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    rc = end_call(c, s);
    return rc ? rc : turn;
}

// The flow field of the most recent ofarn_stream_next_view turn, fetched after the fact (device staging -> host).
int ofarn_stream_view_flow(ofarn_ctx *c, int w, int h, float *h_flow)
{
    if (!c || !h_flow) return fail(OFARN_E_INVALID, "ctx or flow is NULL");
    const ofarn_ctx::Stream &st = c->stream_state;
    if (!(st.have && st.view_flow_valid && st.w == w && st.h == h) || !c->st_flow)
        return fail(OFARN_E_INVALID, "no streaming turn of %dx%d has produced a flow on this context yet", w, h);
    OFARN_ON_DEVICE(c->device);
    int rc = begin_call(c, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h_flow, c->st_flow, (size_t)w * h * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return end_call(c, c->stream);
}

// draw_sparse_lamps for the danger map of the most recent ofarn_stream_next_view turn, which is still on the device; with over_frame
// the layer is cv2.add-ed onto the turn's BGR frame (pathfinder_viewer.py:299-300), which the turn uploaded anyway.
int ofarn_stream_view_lamps(ofarn_ctx *c, int w, int h, int radius, int over_frame, uint8_t *h_out)
{
    if (!c || !h_out) return fail(OFARN_E_INVALID, "ctx or out is NULL");
    ofarn_ctx::Stream &st = c->stream_state;
    if (!(st.have && st.view_danger_valid && st.w == w && st.h == h) || !st.d_view)
        return fail(OFARN_E_INVALID, "no ofarn_stream_next_view turn of %dx%d has produced a danger map on this context yet", w, h);
    if (over_frame && !st.view_bgr_valid)
        return fail(OFARN_E_INVALID, "over_frame needs the turn's frame in BGR, and the most recent view turn was given a gray frame");
    LampGrid g;
    int P = 0;
    int rc = lamp_grid(c, w, h, radius, &g, &P);
    if (rc) return rc;
    if (P != st.view_P) return fail(OFARN_E_INVALID, "the turn's danger map has %d points, the grid of %dx%d has %d", st.view_P, w, h, P);
    OFARN_ON_DEVICE(c->device);
    const size_t img = (size_t)w * h * 3;
    if ((rc = grow_u8(c, &st.d_lamps, &st.lamps_cap, img, "lamp layer"))) return rc;
    hipStream_t s = c->stream;
    if ((rc = begin_call(c, s))) return rc;
    const uint8_t *d_mask = st.d_view + st.view_mask_off;
    launch_draw_lamps(s, d_mask, d_mask + P, P, over_frame ? st.d_bgr : nullptr, st.d_lamps, w, h, 1, g);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_out, st.d_lamps, img, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return end_call(c, s);
}

// draw_flow as an image for the flow the most recent ofarn_stream_next_view turn left on the device: the arrows rasterised as
// cv2.polylines / cv2.circle draw them (DenseOF.py:40-59); with over_frame onto the turn's BGR frame (cv2.add, DenseOF.py:574).
int ofarn_stream_view_arrows(ofarn_ctx *c, int w, int h, int step, int over_frame, uint8_t *h_out)
{
    if (!c || !h_out) return fail(OFARN_E_INVALID, "ctx or out is NULL");
    if (step < 1) return fail(OFARN_E_INVALID, "arrow step must be >= 1, got %d", step);
    ofarn_ctx::Stream &st = c->stream_state;
    if (!(st.have && st.view_flow_valid && st.w == w && st.h == h) || !c->st_flow)
        return fail(OFARN_E_INVALID, "no streaming turn of %dx%d has produced a flow on this context yet", w, h);
    if (over_frame && !st.view_bgr_valid)
        return fail(OFARN_E_INVALID, "over_frame needs the turn's frame in BGR, and the most recent view turn was given a gray frame");
    OFARN_ON_DEVICE(c->device);
    const size_t img = (size_t)w * h * 3;
    int rc = grow_u8(c, &st.d_lamps, &st.lamps_cap, img, "arrow layer");
    if (rc) return rc;
    double astart = 0;
    const int nx = arrow_axis(w, step, &astart), ny = arrow_axis(h, step, &astart);
    hipStream_t s = c->stream;
    if ((rc = begin_call(c, s))) return rc;
    if (over_frame) HIP_TRY(hipMemcpyAsync(st.d_lamps, st.d_bgr, img, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemsetAsync(st.d_lamps, 0, img, s));
    launch_draw_flow(s, c->st_flow, w, h, 1, nx, ny, astart, (double)step, st.d_lamps);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_out, st.d_lamps, img, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return end_call(c, s);
}

// draw_hsv of the flow the most recent ofarn_stream_next_view turn left on the device; with over_frame added onto the turn's BGR frame,
// cv2.add(output_bgr, draw_hsv(flow)) (DenseOF.py:577-578).
int ofarn_stream_view_rainbow(ofarn_ctx *c, int w, int h, int over_frame, uint8_t *h_out)
{
    if (!c || !h_out) return fail(OFARN_E_INVALID, "ctx or out is NULL");
    ofarn_ctx::Stream &st = c->stream_state;
    if (!(st.have && st.view_flow_valid && st.w == w && st.h == h) || !c->st_flow)
        return fail(OFARN_E_INVALID, "no streaming turn of %dx%d has produced a flow on this context yet", w, h);
    if (over_frame && !st.view_bgr_valid)
        return fail(OFARN_E_INVALID, "over_frame needs the turn's frame in BGR, and the most recent view turn was given a gray frame");
    OFARN_ON_DEVICE(c->device);
    const size_t img = (size_t)w * h * 3;
    int rc = grow_u8(c, &st.d_lamps, &st.lamps_cap, img, "rainbow layer");
    if (rc) return rc;
    hipStream_t s = c->stream;
    if ((rc = begin_call(c, s))) return rc;
    launch_flow_hsv(s, c->st_flow, (size_t)w * h, nullptr, st.d_lamps, over_frame ? st.d_bgr : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_out, st.d_lamps, img, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return end_call(c, s);
}

// Pipelined submission: enqueue the turn and return; the flow lands in h_flow asynchronously (copy stream), while the caller
// already submits the next frame -- whose kernels then run beside this turn's device-to-host transfer (16.6 MB at 1080p, about as
// long as the kernels themselves).
int ofarn_stream_submit(ofarn_ctx *c, const uint8_t *h_gray, int w, int h, int stride, float *h_flow)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    OFARN_ON_DEVICE(c->device);
    int rc = stream_check(c, w, h);
    if (rc) return rc;
    if (!h_gray) return fail(OFARN_E_INVALID, "frame is NULL");
    if (stride < w) return fail(OFARN_E_INVALID, "stride %d < width %d", stride, w);
    if (c->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW)
        return fail(OFARN_E_UNSUPPORTED, "ofarn_stream_submit does not take OPTFLOW_USE_INITIAL_FLOW (the flow buffer is still in flight "
                    "when the next turn would need it): use ofarn_stream_next");
    ofarn_ctx::Stream &st = c->stream_state;
    const bool had = st.have;
    if (had && !h_flow) return fail(OFARN_E_INVALID, "flow is NULL");
    const size_t fsz = (size_t)w * h;
    if ((rc = grow_u8(c, &st.d_frame, &st.frame_cap, fsz, "streaming frame buffer"))) return rc;
    if (!st.copy_stream) {
        // Lowest stream priority: streams of one priority share a small pool of hardware queues, and which streams end up on one
        // queue depends on how many the process created before (measured: the pipelined loop takes 0.58, 0.69 or 0.88 ms per
        // frame depending on that alone).  A priority of its own gives the transfer its own queue whatever else exists, and lets
        // the next turn's kernels go first when both have workgroups to place.
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_least = 0; }
        if (hipStreamCreateWithPriority(&st.copy_stream, hipStreamNonBlocking, prio_least) != hipSuccess) {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&st.copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(OFARN_E_HIP, "stream creation failed");
        }
        for (int i = 0; i < ofarn_ctx::Stream::kRing; i++)
            if (hipEventCreateWithFlags(&st.ev_computed[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&st.ev_copied[i], hipEventDisableTiming) != hipSuccess)
                return fail(OFARN_E_HIP, "event creation failed");
        for (int i = 0; i < 2; i++)
            if (hipEventCreateWithFlags(&st.ev_uploaded[i], hipEventDisableTiming) != hipSuccess)
                return fail(OFARN_E_HIP, "event creation failed");
        if (hipEventCreateWithFlags(&st.ev_src_uploaded, hipEventDisableTiming) != hipSuccess)
            return fail(OFARN_E_HIP, "event creation failed");
    }
    // A frame that was page-locked already is uploaded from where it lies: the caller may rewrite that buffer (a capture loop
    // reading every frame into one pinned_empty() array) once ofarn_stream_submit has been called AGAIN or ofarn_stream_wait has
    // returned -- both wait here for the previous turn's upload, which finished long ago in any loop that does work in between.
    if (st.src_uploaded_valid) { HIP_TRY(hipEventSynchronize(st.ev_src_uploaded)); st.src_uploaded_valid = false; }
    if (fsz * 2 > st.ring_cap) {
        HIP_TRY(hipStreamSynchronize(st.copy_stream));            // a transfer may still read the old buffers
        for (int i = 0; i < ofarn_ctx::Stream::kRing; i++) {
            if (st.ring[i]) { (void)hipFree(st.ring[i]); c->ws_bytes -= st.ring_cap * sizeof(float) + 256; st.ring[i] = nullptr; }
            st.copied_valid[i] = false;
        }
        st.ring_cap = 0;
        for (int i = 0; i < ofarn_ctx::Stream::kRing; i++) {
            if (hipMalloc((void **)&st.ring[i], fsz * 2 * sizeof(float) + 256) != hipSuccess) {
                (void)hipGetLastError();
                st.ring[i] = nullptr;
                return fail(OFARN_E_NOMEM, "flow ring buffer of %zu bytes does not fit", fsz * 2 * sizeof(float));
            }
            c->ws_bytes += fsz * 2 * sizeof(float) + 256;
        }
        st.ring_cap = fsz * 2;
    }
    hipStream_t s = c->stream;
    if ((rc = begin_call(c, s))) return rc;
    const int slot = (int)(st.submits % ofarn_ctx::Stream::kRing);
    // the transfer of the turn kRing turns ago read ring[slot]: the kernels that overwrite it wait for that transfer
    if (st.copied_valid[slot]) HIP_TRY(hipStreamWaitEvent(s, st.ev_copied[slot], 0));
    {
        // a frame in pageable memory goes through one of two page-locked staging buffers (a host copy of w x h bytes): the upload is
        // then truly asynchronous, the call does not wait for the previous turn's kernels, and the caller's array is free again
        // when the call returns; a frame that already is page-locked is uploaded from where it lies
        hipPointerAttribute_t at;
        const bool pinned = hipPointerGetAttributes(&at, h_gray) == hipSuccess && at.type == hipMemoryTypeHost;
        if (!pinned) (void)hipGetLastError();
        const uint8_t *src = h_gray;
        int sstride = stride;
        int k = -1;
        if (!pinned) {
            k = (int)(st.stages++ & 1);
            if (fsz > st.stage_cap) {
                for (int i = 0; i < 2; i++) {
                    if (st.uploaded_valid[i]) { HIP_TRY(hipEventSynchronize(st.ev_uploaded[i])); st.uploaded_valid[i] = false; }
                    if (st.h_stage[i]) { (void)hipHostFree(st.h_stage[i]); st.h_stage[i] = nullptr; }
                }
                st.stage_cap = 0;
                for (int i = 0; i < 2; i++)
                    if (hipHostMalloc((void **)&st.h_stage[i], fsz, hipHostMallocDefault) != hipSuccess) {
                        (void)hipGetLastError();
                        st.h_stage[i] = nullptr;
                        (void)end_call(c, s);
                        return fail(OFARN_E_NOMEM, "page-locked frame staging of %zu bytes could not be allocated", fsz);
                    }
                st.stage_cap = fsz;
            }
            if (st.uploaded_valid[k]) HIP_TRY(hipEventSynchronize(st.ev_uploaded[k]));     // its previous upload (two turns ago) is long done
            for (int y = 0; y < h; y++) memcpy(st.h_stage[k] + (size_t)y * w, h_gray + (size_t)y * stride, (size_t)w);
            src = st.h_stage[k];
            sstride = w;
        }
        if (sstride == w) HIP_TRY(hipMemcpyAsync(st.d_frame, src, fsz, hipMemcpyHostToDevice, s));
        else HIP_TRY(hipMemcpy2DAsync(st.d_frame, w, src, sstride, w, h, hipMemcpyHostToDevice, s));
        if (k >= 0) { HIP_TRY(hipEventRecord(st.ev_uploaded[k], s)); st.uploaded_valid[k] = true; }
        else { HIP_TRY(hipEventRecord(st.ev_src_uploaded, s)); st.src_uploaded_valid = true; }
    }
    // No side stream for stages A + B here: the previous turn's transfer is running on the copy stream, and a THIRD hardware queue in
    // play means that -- depending on which queues the driver happens to put on one pipe -- the level builds can sit behind the
    // 0.3 ms blit kernel while the iteration chain waits for them (measured: 1.0-1.6 ms per frame instead of 0.58 for two of eight
    // stream-creation histories).  The overlap is worth 20 us; not here.
    const int keep_overlap = c->stream_overlap;
    c->stream_overlap = 0;
    const int turn = stream_turn(c, s, st.d_frame, w, h, st.ring[slot], nullptr, nullptr);
    c->stream_overlap = keep_overlap;
    if (turn < 0) { (void)end_call(c, s); return turn; }
    if (turn == OFARN_OK) {
        HIP_TRY(hipEventRecord(st.ev_computed[slot], s));
        HIP_TRY(hipStreamWaitEvent(st.copy_stream, st.ev_computed[slot], 0));
        // page-locked destination: pushed by a small kernel that leaves the CUs to the next turn (see k_push_host); else a plain copy
        void *mapped = nullptr;
        if (c->push_blocks > 0 && (fsz * 2) % 4 == 0 && ((uintptr_t)h_flow & 15) == 0)
            mapped = mapped_host_range(h_flow, fsz * 2 * sizeof(float));
        if (mapped) launch_push_host(st.copy_stream, st.ring[slot], static_cast<float *>(mapped), fsz * 2, c->push_blocks);
        else HIP_TRY(hipMemcpyAsync(h_flow, st.ring[slot], fsz * 2 * sizeof(float), hipMemcpyDeviceToHost, st.copy_stream));
        HIP_TRY(hipEventRecord(st.ev_copied[slot], st.copy_stream));
        st.copied_valid[slot] = true;
        st.submits++;
    }
    rc = end_call(c, s);
    return rc ? rc : turn;
}

int ofarn_stream_wait(ofarn_ctx *c, int leave_in_flight)
{
    if (!c) return fail(OFARN_E_INVALID, "ctx is NULL");
    constexpr int K = ofarn_ctx::Stream::kRing;
    if (leave_in_flight < 0 || leave_in_flight >= K) return fail(OFARN_E_INVALID, "leave_in_flight must be in [0, %d]", K - 1);
    OFARN_ON_DEVICE(c->device);
    ofarn_ctx::Stream &st = c->stream_state;
    if (st.src_uploaded_valid) { HIP_TRY(hipEventSynchronize(st.ev_src_uploaded)); st.src_uploaded_valid = false; }
    if (leave_in_flight > 0) {
        // everything but the `leave_in_flight` most recent turns: turn n (0-based) used slot n % K; transfers complete in order, so
        // waiting for the newest turn that must be complete is enough
        if (st.submits > (unsigned long long)leave_in_flight) {
            const int slot = (int)((st.submits - 1 - leave_in_flight) % K);
            if (st.copied_valid[slot]) HIP_TRY(hipEventSynchronize(st.ev_copied[slot]));
        }
        return OFARN_OK;
    }
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    if (st.copy_stream) HIP_TRY(hipStreamSynchronize(st.copy_stream));
    return OFARN_OK;
}

int ofarn_host_alloc(size_t bytes, void **out)
{
    if (!out) return fail(OFARN_E_INVALID, "out is NULL");
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(OFARN_E_NOMEM, "pinned host buffer of %zu bytes could not be allocated", bytes);
    }
    std::lock_guard<std::mutex> lk(g_pins_mu);
    g_pins[(uintptr_t)*out] = bytes ? bytes : 1;
    return OFARN_OK;
}

int ofarn_host_free(void *p)
{
    if (!p) return OFARN_OK;
    {
        std::lock_guard<std::mutex> lk(g_pins_mu);
        g_pins.erase((uintptr_t)p);
    }
    HIP_TRY(hipHostFree(p));
    return OFARN_OK;
}

#pragma GCC visibility pop
}  // extern "C"
