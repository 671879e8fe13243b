// ofarn_host.h -- host-side internals shared by the C-ABI translation units (ofarn_api*.hip): the context,
// the error helper, per-call device scratch, per-kernel timing and the plan / workspace helpers of ofarn_api.hip.
#pragma once
#include "../../include/ofarn.h"
#include "ofarn_internal.h"

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ofarn_host {

// cv2.cvtColor(COLOR_BGR2GRAY) coefficients: color_rgb.simd.hpp RGB2Gray<uchar>, 15-bit fixed point (B, G, R)
constexpr int kGrayB = 3735, kGrayG = 19235, kGrayR = 9798, kGrayShift = 15;

// thread-local message of the last failing call (ofarn_last_error); returns `code`
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(OFARN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                  \
    } while (0)

// Every entry point runs on its context's device and leaves the caller's current device as it found it (a host application with
// its own device state must not notice the library).  OFARN_ON_DEVICE(dev) at the top of an entry point: switches if needed,
// switches back when the entry point returns -- on every path.  Helpers called from entry points use plain hipSetDevice.
// Test hook (one GPU boxes cannot observe a switch): ofarn_debug_device_scope() makes the scope believe the caller was on another
// ordinal and reports the ordinal it restored.
extern int g_fake_current_device;              // -1: ask hipGetDevice
extern thread_local int t_last_restored_device, t_device_scopes;
struct DeviceScope {
    int before = -1;
    bool restore = false, fake = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        t_device_scopes++;
        if (g_fake_current_device >= 0) { before = g_fake_current_device; fake = true; }
        else if (hipGetDevice(&before) != hipSuccess) { (void)hipGetLastError(); before = -1; }
        if (before != device) { err = hipSetDevice(device); restore = before >= 0; }
    }
    ~DeviceScope()
    {
        if (!restore) return;
        t_last_restored_device = before;
        if (!fake) (void)hipSetDevice(before);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define OFARN_ON_DEVICE(device)                                                                              \
    ofarn_host::DeviceScope device_scope_(device);                                                           \
    if (device_scope_.err != hipSuccess)                                                                     \
        return ofarn_host::fail(OFARN_E_HIP, "hipSetDevice(%d) failed: %s", (int)(device), hipGetErrorString(device_scope_.err))

// device scratch that lives for one host-pointer call
struct DevTmp {
    void *p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(OFARN_E_NOMEM, "device scratch of %zu bytes does not fit", bytes);
        }
        return OFARN_OK;
    }
    template <typename T> T *as() { return static_cast<T *>(p); }
};

struct Level {
    int w = 0, h = 0, ksize = 0;
    double sigma = 0;
    // device tables
    float *d_kern = nullptr;
    float h_kern3[3] = {0, 0, 0};   // host copy of the taps when ksize == 3
    std::vector<float> h_kern;      // host copy of all taps (passed by value to k_level_direct)
    int *d_xofs = nullptr, *d_yofs = nullptr;        // image resize W->w, H->h
    float *d_xa = nullptr, *d_ya = nullptr;
    int *d_fxofs = nullptr, *d_fyofs = nullptr;      // flow resize (k+1) -> k
    float *d_fxa = nullptr, *d_fya = nullptr;
};

}  // namespace ofarn_host

struct ofarn_ctx {
    ofarn_params prm{};
    int device = 0;
    int max_w = 0, max_h = 0, max_batch = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ofarn::PolyCoef poly{};
    float *d_gwin = nullptr;     // m+1 taps of the OPTFLOW_FARNEBACK_GAUSSIAN window
    std::vector<float> h_gwin;   // host copy (passed by value to the fused Gaussian iteration kernel)
    // plan (cached for one frame size)
    int plan_w = 0, plan_h = 0;
    std::vector<ofarn_host::Level> lv;
    int *d_pts = nullptr;
    int P = 0;
    ofarn::AreaTabHost area;            // resize(INTER_AREA) full size -> coarsest level (OPTFLOW_USE_INITIAL_FLOW)
    float init_scale = 1.f;      // pyr_scale ^ levels, as optflowgf.cpp accumulates it
    std::vector<void *> plan_allocs;
    // workspace
    // two workspaces: waves of one batch alternate between them on two internal streams, so the tail
    // of one wave's kernels overlaps the other wave's (the second is allocated on first use)
    // R and the two flow buffers are sized for max_batch pairs when the context is created; tmp, I and M grow on
    // demand to what the schedule of the call at hand needs (the fused default path needs no M and, at 1080p,
    // 1.6 MB of tmp and 2 MB of I per frame instead of the 16.6 + 8.3 MB a full-resolution buffer would take).
    struct Workspace {
        float *tmp = nullptr, *I = nullptr, *R = nullptr, *M = nullptr, *flowA = nullptr, *flowB = nullptr;
        size_t cap_tmp = 0, cap_I = 0, cap_R = 0, cap_M = 0, cap_flow = 0;   // capacities in floats
    };
    Workspace ws[2];
    // Ordering between calls on different streams: every entry point that enqueues work on the shared workspace
    // records ev_done behind it; the next call waits for that event on ITS stream when the stream differs.
    hipEvent_t ev_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    bool dual = true;            // OFARN_SINGLE_STREAM=1 disables the second workspace
    uint64_t ws_bytes = 0;
    uint8_t *gray[2] = {nullptr, nullptr};   // gray frames of a wave when the caller hands over BGR (lazy)
    // sparse LK (lazy): pyramid levels >= 1 (uint8) and Scharr derivatives of every level (int16 x 2) for one wave of frames
    struct LkWs {
        int w = 0, h = 0, frames = 0, levels = -1;
        std::vector<int> lw, lh;
        std::vector<uint8_t *> pyr;      // pyr[0] unused (level 0 is the caller's frames)
        std::vector<int16_t *> der;
    } lk;
    // Streaming session (ofarn_stream_*, ofarn_api_stream.hip): the reference's frame loop hands over ONE new frame per turn
    // and carries the previous one (DenseOF.py:510, 519-525: prev_gray = gray).  The polynomial expansion of the last frame
    // is kept for every pyramid level in one of two slots; the next frame's goes into the other slot and the iteration
    // kernels read the pair as (slot cur, the other slot) -- see pair_frames() for the negative frame step.
    struct Stream {
        int w = 0, h = 0;
        bool have = false;              // slot `cur` holds a frame
        int cur = 0;
        float *R = nullptr;             // level k: two slots of r_frame_stride(w_k * h_k) floats at R + off[k]
        size_t cap = 0;                 // floats
        std::vector<size_t> off;
        uint8_t *d_frame = nullptr;     // the new gray frame (host entry points / BGR input)
        uint8_t *d_bgr = nullptr;
        size_t frame_cap = 0, bgr_cap = 0;
        uint8_t *d_view = nullptr;      // ofarn_stream_next_view: arrow lines + rainbow image of the turn, on the device
        size_t view_cap = 0;
        uint8_t *h_view = nullptr;      // page-locked landing zone of the small results (mask, v, lines): one transfer, then host copies
        size_t h_view_cap = 0;
        bool view_flow_valid = false;   // c->st_flow holds the flow of the most recent turn (ofarn_stream_next_view)
        bool view_danger_valid = false; // d_view + view_mask_off holds mask[view_P], v[view_P] of the most recent turn (ofarn_stream_view_lamps)
        bool view_bgr_valid = false;    // d_bgr holds the BGR frame of that turn
        size_t view_mask_off = 0;
        int view_P = 0;
        uint8_t *d_lamps = nullptr;     // ofarn_stream_view_lamps: the lamp layer / composited frame on the device
        size_t lamps_cap = 0;
        unsigned long long turns = 0;
        // pipelined submission (ofarn_stream_submit / ofarn_stream_wait): two device flow buffers used in turn, a copy stream whose
        // device-to-host transfer of turn t runs beside the kernels of turn t+1
        static constexpr int kRing = 3;  // turns whose flow can be on its way to the host at once + the one being computed
        float *ring[kRing] = {nullptr, nullptr, nullptr};
        size_t ring_cap = 0;            // floats, each
        hipStream_t copy_stream = nullptr;
        hipEvent_t ev_computed[kRing] = {nullptr, nullptr, nullptr}, ev_copied[kRing] = {nullptr, nullptr, nullptr};
        bool copied_valid[kRing] = {false, false, false};
        unsigned long long submits = 0;
        // page-locked staging of the submitted frames: an asynchronous upload straight from pageable memory would make the
        // host wait for the stream (the previous turn's kernels) inside the submit call
        uint8_t *h_stage[2] = {nullptr, nullptr};
        size_t stage_cap = 0;
        hipEvent_t ev_uploaded[2] = {nullptr, nullptr};
        bool uploaded_valid[2] = {false, false};
        unsigned long long stages = 0;
        hipEvent_t ev_src_uploaded = nullptr;   // behind the upload of a frame that was page-locked already (read in place): the next
        bool src_uploaded_valid = false;        // submit / wait makes sure it has been read before the caller may rewrite it
        // ofarn_calc_reuse: byte copies of the frames the session was last given, page-locked (they double as upload staging);
        // h_keep[keep_cur] equals the frame in R slot `cur` iff keep_valid && keep_turn == turns
        uint8_t *h_keep[2] = {nullptr, nullptr};
        size_t keep_cap = 0;
        int keep_cur = 0;
        bool keep_valid = false;
        unsigned long long keep_turn = 0, reuse_hits = 0, reuse_misses = 0;
    } stream_state;
    int stream_overlap = 2;             // streaming turn: stages A + B on an internal stream beside the iteration chain (run_wave);
                                        // 1: the chain waits behind every level's expansion, 2: behind the coarsest and every second one
    hipEvent_t ev_level[32] = {nullptr};
    int push_blocks = 0;                // ofarn_stream_submit: > 0 pushes a finished flow field to pinned host memory with a kernel of that many
                                        // blocks instead of hipMemcpyAsync ("push_blocks"; measured slower at every size, kept as an experiment)
    int stream_zero_copy = 1;           // ofarn_stream_next: let the last kernel write a pinned flow buffer itself (OFARN_STREAM_ZERO_COPY=0: copy)
    // host-API staging (lazy)
    uint8_t *st_frames = nullptr;
    float *st_flow = nullptr;
    uint8_t *st_mask = nullptr, *st_v = nullptr;
    size_t st_frames_cap = 0, st_flow_cap = 0, st_dm_cap = 0;
    double last_ms = 0;
    // per-kernel profiling (ofarn_profile_*): hipEvent pairs around each launch, on the launch stream
    struct ProfRec { int stage, level; double units; hipEvent_t a, b; };
    bool prof_on = false;
    bool prof_dual = false;       // "prof_dual": per-kernel timing does NOT force the waves of a batch onto one stream (each launch is
                                  // bracketed on the stream it runs on; durations then include what co-running kernels take from it)
    bool box_running = false;     // "box_order" = 1: the box window summed in OpenCV's literal order (k_vsum_running + k_hsum_running_solve,
                                  // oracle OFO_BOX_RUNNING) instead of the restarted sums of the throughput kernels; unfused, 3-4 x slower
    bool force_generic = false;   // OFARN_FORCE_GENERIC=1 or ofarn_set_option: use the unfused kernels only
    int debug_fail_wave = -1;     // test hook: the (n+1)-th wave from now returns OFARN_E_NOMEM (error-path tests); -1 = off
    int tile_mode = -1;           // fused iteration: -1 = tile kernel for small grids, marching kernel otherwise; 0 / 1 = never / always
                                  // the tile kernel (OFARN_TILE when the context is created, ofarn_set_option "tile")
    int row_small_symm = 1;       // GaussianBlur row pass of a 3- or 5-tap kernel in SymmRowSmallFilter's order (oracle
                                  // OFO_ROW_SMALL_SYMM); OFARN_ROW_LTR=1 when the context is created: left to right (rounds 1-2)
    int direct_min_frames = 16;   // k_level_direct marches long strips per thread: below this many frames of 1920 x 1080 (by pixel
                                  // count: 4 frames of 3840 x 2160) in a wave the row-pass + column-pass pair has more parallelism
                                  // and lower latency (OFARN_DIRECT_MIN_FRAMES)
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> prof_free;
    struct ProfAcc { int launches = 0; double ms = 0, units = 0; };
    ProfAcc prof_acc[OFARN_STAGE_COUNT][32];
};

namespace ofarn_host {

// hip_stream argument of the C-ABI: NULL = the context's own stream, OFARN_STREAM_NULL = HIP's null (legacy default) stream,
// whose handle 0 could not be told from NULL otherwise; anything else is a hipStream_t.
inline hipStream_t pick_stream(const ofarn_ctx *c, void *hip_stream)
{
    if (hip_stream == OFARN_STREAM_NULL) return nullptr;
    return hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
}

// ofarn_api_stream.hip: the device-side address of a page-locked host range [p, p + bytes), or nullptr when p is not page-locked
// or the allocation it lies in does not cover the whole range (the caller then takes the copy path).
void *mapped_host_range(const void *p, size_t bytes);
// ofarn_api_extras.hip: np.mgrid[step/2:size:step] of draw_flow (DenseOF.py:44): count and float start
int arrow_axis(int size, int step, double *start);
// Lamp layer geometry for w x h frames on the context's measurement grid; OFARN_E_* if the discs would touch or the radius is out of range
int lamp_grid(const ofarn_ctx *c, int w, int h, int radius, ofarn::LampGrid *g, int *P);
// ofarn_api.hip
int check_size(ofarn_ctx *c, int w, int h);
int axis_points(int size, int step, std::vector<int> *out);   // the measurement grid along one axis (pathfinder_viewer.py:255-263)
int make_plan(ofarn_ctx *c, int w, int h);
int ensure_staging(ofarn_ctx *c, size_t frames_bytes, size_t flow_bytes, size_t dm_bytes);
int build_area_tab(ofarn_ctx *c, int sw, int sh, int dw, int dh, ofarn::AreaTabHost &out);
void resize_tables(int ssize, int dsize, std::vector<int> &ofs, std::vector<float> &alpha);
hipEvent_t prof_event(ofarn_ctx *c);   // nullptr if hipEventCreate fails (the launch then goes untimed)
// Grow-only reservation of workspace `wi`, sizes in floats (0 = leave alone).  OFARN_E_NOMEM if it does not fit; with
// `capturing` a buffer that would have to grow is an error instead (growing frees and allocates: not capturable, and a graph
// captured earlier would keep replaying into the freed buffer).
int ws_reserve(ofarn_ctx *c, int wi, size_t tmp, size_t I, size_t R, size_t M, size_t flow, bool capturing = false);
bool stream_is_capturing(hipStream_t s);
// Call ordering on the shared workspace (see ofarn_ctx::ev_done): begin_call before the first enqueue of an entry
// point on stream s, end_call behind its last one.
int begin_call(ofarn_ctx *c, hipStream_t s);
int end_call(ofarn_ctx *c, hipStream_t s);
// One wave of the dense schedule (ofarn_api.hip).  With `st` (streaming turn): d_frames is the ONE new frame, its level images and
// polynomial expansions go to the free slot of st->R, and -- if a previous frame is held -- the pair (previous, new) is iterated.
int run_wave(ofarn_ctx *c, hipStream_t s, const uint8_t *d_frames, int npairs, int pairs_mode, int w, int h, float *d_flow,
             uint8_t *d_mask, uint8_t *d_v, int wi = 0, const float *d_init = nullptr, ofarn_ctx::Stream *st = nullptr,
             bool st_pair = false);

// Runs `launch` and, when profiling is on, brackets it with two events on the same stream.
template <typename F>
inline void timed(ofarn_ctx *c, hipStream_t s, int stage, int level, double units, F &&launch)
{
    if (!c->prof_on) { launch(); return; }
    ofarn_ctx::ProfRec r{stage, level, units, prof_event(c), prof_event(c)};
    if (!r.a || !r.b) {                       // no event to be had: run the launch untimed rather than record on a null handle
        if (r.a) c->prof_free.push_back(r.a);
        if (r.b) c->prof_free.push_back(r.b);
        launch();
        return;
    }
    (void)hipEventRecord(r.a, s);
    launch();
    (void)hipEventRecord(r.b, s);
    c->prof_pending.push_back(r);
}

}  // namespace ofarn_host
