/*
 * ofarn.h -- C-ABI of libofarn.so: dense Farneback optical flow + grid vector filter on MI355X.
 *
 * This is the drop-in boundary for the one hot path of spirinis/HackathonOpticalFlow:
 *
 *   calculate_optical_flow(prev, next, flow=None, pyr_scale=0.5, levels=3, winsize=15,
 *                          iterations=3, poly_n=5, poly_sigma=1.2, flags=0) -> float32[H,W,2]
 *       reference: DenseOF.py:127-157, which forwards to cv2.calcOpticalFlowFarneback
 *       (DenseOF.py:147-156); called once per frame at DenseOF.py:520.
 *
 * followed by the reference's own sparse-grid vector filter and danger brightness
 *       reference: pathfinder_viewer.py:159-176 (filter), :204-217 (V), :252-267 (grid).
 *
 * The reference is pure Python over the cv2 wheel, so it has no FFI of its own for this path; the
 * entry points below are what a ctypes binding in DenseOF.py would call instead of cv2 (the stub
 * is shown in INTEGRATION.md).  Plain pointers and sizes only; no C++ or torch types; no
 * exceptions cross this boundary.  Every function returns OFARN_OK (0) or a negative error code
 * and leaves a message for ofarn_last_error() (thread local).
 *
 * THE CONTRACT is seven entry points: ofarn_create, ofarn_calc, ofarn_calc_batch (+ _device), ofarn_grid_filter,
 * ofarn_destroy, ofarn_last_error and the timing getter ofarn_last_device_ms.  Everything else in this header is one of
 *   - the same path in the shape the reference's frame loop uses it (ofarn_stream_*, ofarn_calc_reuse) or across GPUs (ofarn_multi_*),
 *   - the "next" rows of SURVEY 8(f) (front end, flags, visualisers, sparse LK),
 *   - per-stage entry points and switches for the parity tests and measurements,
 *   - extras outside the contract, grouped at the END of the file under their own banner.
 *
 * Pointer naming: h_* = host memory, d_* = device (HBM) memory of the context's GPU.
 * A context is bound to one GPU and must not be used from two threads at once.
 *
 * Streams and ordering.  Device entry points (*_device) enqueue on the caller's `hip_stream` and return without
 * synchronising; host entry points run on the context's own stream and return when their result is in host memory.
 * All entry points of one context share its workspace, so the context orders its calls itself: every call records an
 * event behind its last kernel, and the next call -- whatever stream it is given -- first makes that stream wait for
 * the event.  Calls issued from one thread therefore behave as if the context had a single queue; work the CALLER has
 * queued on other streams (producing the input frames, consuming the flow) is the caller's to order, as with any
 * stream-based library.
 */
#ifndef OFARN_H
#define OFARN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFARN_OK 0
#define OFARN_E_INVALID (-1)   /* bad argument (maps to cv2.error / ValueError)                 */
#define OFARN_E_UNSUPPORTED (-2) /* outside what is built: unknown flag bits, LK windows wider than 64 */
#define OFARN_E_HIP (-3)       /* HIP runtime failure                                            */
#define OFARN_E_NOMEM (-4)     /* workspace does not fit                                         */
#define OFARN_E_SIZE (-5)      /* frame or batch larger than the context was created for         */

/* flags (cv2 names OPTFLOW_*).  With USE_INITIAL_FLOW the flow buffer of every calc entry point is an
 * in/out argument, as cv2's `flow` is: on entry it holds the full-resolution initial flow of each pair. */
#define OFARN_FLAG_USE_INITIAL_FLOW 4
#define OFARN_FLAG_FARNEBACK_GAUSSIAN 256   /* winsize 6..17: fused marching kernel; wider windows: one kernel per stage */

/* pairs_mode of the batch entry points */
#define OFARN_PAIRS_INDEPENDENT 0 /* frames (2i, 2i+1) form pair i; n_pairs = n_frames/2          */
#define OFARN_PAIRS_CONSECUTIVE 1 /* frames (i, i+1) form pair i (video order, DenseOF.py:525: */
                                  /* prev_gray = gray); n_pairs = n_frames-1                     */

/* The seven keyword arguments of DenseOF.py:127-128 (== cv2.calcOpticalFlowFarneback) plus the
 * measurement-grid step of pathfinder_viewer.py:16. */
typedef struct ofarn_params {
    double pyr_scale;   /* < 1                                        default 0.5 */
    int levels;         /* pyramid reductions; levels+1 scales run    default 3   */
    int winsize;        /* averaging window, >= 2                     default 15  */
    int iterations;     /* per level                                  default 3   */
    int poly_n;         /* polynomial-expansion radius (2n+1 taps)    default 5   */
    double poly_sigma;  /*                                            default 1.2 */
    int flags;          /* 0 or an OR of OFARN_FLAG_*                 default 0   */
    int grid_step;      /* danger-map grid step in pixels             default 30  */
    int filter_variant; /* 0: pathfinder_viewer.py:173  median < mod < P99   default 0   */
                        /* 1: DenseOF.py:228            mod > median * 1.2               */
} ofarn_params;

typedef struct ofarn_ctx ofarn_ctx;

/* Fills *p with the reference defaults (DenseOF.py:127-128, pathfinder_viewer.py:16). */
void ofarn_default_params(ofarn_params *p);

/* Creates a context on GPU `device` able to process frames up to max_w x max_h, in waves of up
 * to max_batch pairs resident at once.  Replaces: constructing cv::FarnebackOpticalFlow inside
 * cv2.calcOpticalFlowFarneback (DenseOF.py:147). */
int ofarn_create(const ofarn_params *params, int device, int max_w, int max_h, int max_batch,
                 ofarn_ctx **out);
void ofarn_destroy(ofarn_ctx *ctx);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *ofarn_last_error(void);

/* One frame pair, host memory in, host memory out; synchronous.
 * Replaces: cv2.calcOpticalFlowFarneback(prev, next, None, ...) at DenseOF.py:147-156.
 * h_prev/h_next: uint8, `stride` bytes per row.  h_flow: float32[h][w][2], (dx, dy) per pixel; a page-locked buffer
 * (ofarn_host_alloc) is written by the last kernel itself, without a copy behind it. */
int ofarn_calc(ofarn_ctx *ctx, const uint8_t *h_prev, const uint8_t *h_next, int w, int h,
               int stride, float *h_flow);

/* ofarn_calc for a caller that passes consecutive pairs, as the reference's loop does (DenseOF.py:519-525:
 *     flow = calculate_optical_flow(prev=prev_gray, next=gray); prev_gray = gray),
 * with ofarn_calc's contract: the result depends on (prev, next) alone and equals ofarn_calc's bit for bit for EVERY input.
 * The context keeps a page-locked byte copy of the frame it was last given as `next` (it doubles as the upload's staging
 * buffer).  When all w x h bytes of `prev` equal that copy, the frame already on the device -- with its level images and
 * polynomial expansions -- is reused and only `next` is uploaded and expanded; otherwise both are.  The comparison is a full
 * memcmp on the host, run while the device already computes the turn it would allow (redone from both frames if it fails).
 * Rows of prev / next are stride_prev / stride_next bytes apart.  *reused (may be NULL): 1 if the held frame was reused.
 * h_flow as in ofarn_calc (in/out with OPTFLOW_USE_INITIAL_FLOW; a page-locked buffer is written by the last kernel). */
int ofarn_calc_reuse(ofarn_ctx *ctx, const uint8_t *h_prev, const uint8_t *h_next, int w, int h, int stride_prev,
                     int stride_next, float *h_flow, int *reused);
/* How often ofarn_calc_reuse reused the held frame / started over from both frames on this context (either may be NULL). */
int ofarn_calc_reuse_info(const ofarn_ctx *ctx, unsigned long long *hits, unsigned long long *misses);

/* A batch of frames, host memory; synchronous.  h_frames: uint8[n_frames][h][w] dense.
 * Any of h_flow (float32[n_pairs][h][w][2]), h_mask, h_v (uint8[n_pairs][P]) may be NULL.
 * Replaces: the per-frame loop DenseOF.py:491-525 plus the filter pathfinder_viewer.py:159-176
 * and the V values of pathfinder_viewer.py:204-217 applied to the dense flow sampled at the grid. */
int ofarn_calc_batch(ofarn_ctx *ctx, const uint8_t *h_frames, int n_frames, int w, int h,
                     int pairs_mode, float *h_flow, uint8_t *h_mask, uint8_t *h_v);

/* Same, device-resident: all pointers are HBM addresses on the context's GPU; work is enqueued on
 * `hip_stream` and NOT synchronised.  `hip_stream` is a hipStream_t; NULL = the context's own non-blocking
 * stream; HIP's null (legacy default) stream, whose handle is also 0, is named by OFARN_STREAM_NULL.  The
 * same holds for every `hip_stream` argument below. */
#define OFARN_STREAM_NULL ((void *)(intptr_t)-1)
int ofarn_calc_batch_device(ofarn_ctx *ctx, const uint8_t *d_frames, int n_frames, int w, int h,
                            int pairs_mode, float *d_flow, uint8_t *d_mask, uint8_t *d_v,
                            void *hip_stream);

/* ---- streaming session: the reference's frame loop ---------------------------------
 *     gray = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)                          DenseOF.py:510
 *     flow = calculate_optical_flow(prev=prev_gray, next=gray)               DenseOF.py:519-520
 *     prev_gray = gray                                                       DenseOF.py:525
 * One NEW frame per call.  The context keeps the previous frame's polynomial expansions of every pyramid level on the
 * device, so a turn uploads one frame and runs the level build + polynomial expansion once; the result equals
 * ofarn_calc(previous frame, new frame) bit for bit.  The first frame of a session (after ofarn_create, ofarn_stream_reset or a
 * change of frame size) has nothing to be paired with: the call stores it and returns OFARN_STREAM_PRIMED (> 0) without
 * touching the outputs.  Other entry points of the same context (ofarn_calc, ...) may be called in between; they do not
 * disturb the session.  With OPTFLOW_USE_INITIAL_FLOW the flow buffer is in/out as in ofarn_calc. */
#define OFARN_STREAM_PRIMED 1
/* host frame in, host flow out; synchronous.  h_flow float32[h][w][2] (ignored by the priming call).  A flow buffer from
 * ofarn_host_alloc (pinned) is written by the last kernel directly, without a device-to-host copy behind it. */
int ofarn_stream_next(ofarn_ctx *ctx, const uint8_t *h_gray, int w, int h, int stride, float *h_flow);
/* the same from a packed BGR frame (3 bytes per pixel, `stride` bytes per row): replaces DenseOF.py:510 + :520 */
int ofarn_stream_next_bgr(ofarn_ctx *ctx, const uint8_t *h_bgr, int w, int h, int stride, float *h_flow);
/* the same with the danger map of the pair (uint8[P] each, together or both NULL; pathfinder_viewer.py:159-176, 204-217) */
int ofarn_stream_next_danger(ofarn_ctx *ctx, const uint8_t *h_gray, int w, int h, int stride, float *h_flow,
                             uint8_t *h_mask, uint8_t *h_v);
/* device-resident: frame, flow and danger maps are HBM addresses; enqueued on hip_stream, not synchronised.  d_flow may be NULL
 * when only the danger maps are wanted. */
int ofarn_stream_next_device(ofarn_ctx *ctx, const uint8_t *d_gray, int w, int h, float *d_flow, uint8_t *d_mask,
                             uint8_t *d_v, void *hip_stream);
int ofarn_stream_next_device_bgr(ofarn_ctx *ctx, const uint8_t *d_bgr, int w, int h, float *d_flow, uint8_t *d_mask,
                                 uint8_t *d_v, void *hip_stream);
/* The loop's per-frame OUTPUTS without the flow field crossing PCIe.  What the reference does with `flow` each turn is draw it:
 * draw_flow's arrows on a step-14 grid (DenseOF.py:40-49, :574), draw_hsv's rainbow (DenseOF.py:109-124, :578), the danger points of
 * the grid filter (pathfinder_viewer.py:159-176, 204-217).  This turn returns exactly those -- h_mask / h_v uint8[P] (together or both
 * NULL), h_lines int32[K][2][2] with K = ofarn_flow_arrow_count(w, h, arrow_step) (or NULL), h_rainbow uint8[h][w][3] BGR (or NULL)
 * -- and keeps the float32[h][w][2] flow in HBM; ofarn_stream_view_flow fetches it afterwards if it is wanted after all.
 * h_frame: gray (bgr = 0) or packed BGR (bgr != 0), `stride` bytes per row.  Returns as ofarn_stream_next. */
int ofarn_stream_next_view(ofarn_ctx *ctx, const uint8_t *h_frame, int bgr, int w, int h, int stride, uint8_t *h_mask,
                           uint8_t *h_v, int arrow_step, int32_t *h_lines, uint8_t *h_rainbow);
int ofarn_stream_view_flow(ofarn_ctx *ctx, int w, int h, float *h_flow);
/* Pipelined form for throughput: ofarn_stream_submit enqueues the turn (upload, kernels, transfer of the flow into h_flow on a copy
 * stream) and returns without waiting; the caller submits the next frame at once, whose kernels then run BESIDE this turn's
 * device-to-host transfer (at 1080p the 16.6 MB of flow take about as long over PCIe as the kernels).  h_flow of a turn is complete
 * when ofarn_stream_wait returns for it (leave_in_flight = 0: everything submitted so far; 1 or 2: everything but the most recent
 * one or two turns, which keep running -- the steady state of a pipelined loop); until then it must stay allocated and untouched --
 * page-locked memory (ofarn_host_alloc) makes the transfer truly asynchronous.  h_gray in pageable memory is copied into a
 * page-locked staging buffer by the call and is free again when it returns; a page-locked h_gray is uploaded from where it lies and
 * must stay untouched until the NEXT ofarn_stream_submit / ofarn_stream_wait on this context returns.  Return values as
 * ofarn_stream_next.  Not with
 * OPTFLOW_USE_INITIAL_FLOW (OFARN_E_UNSUPPORTED). */
int ofarn_stream_submit(ofarn_ctx *ctx, const uint8_t *h_gray, int w, int h, int stride, float *h_flow);
int ofarn_stream_wait(ofarn_ctx *ctx, int leave_in_flight);
/* Forgets the held frame: the next call primes again (a cut in the video, a seek: DenseOF.py:476-481 re-reads prev_gray). */
int ofarn_stream_reset(ofarn_ctx *ctx);
/* 1 if the session holds a frame of this size (the next ofarn_stream_next* call will produce a flow), else 0. */
int ofarn_stream_primed(const ofarn_ctx *ctx, int w, int h);
/* Page-locked host memory for frames and flow fields (hipHostMalloc): transfers to and from it need no staging copy.  A flow
 * buffer is written by the GPU in place only when the page-locked allocation covers all of [h_flow, h_flow + 8 w h): blocks from
 * ofarn_host_alloc are checked against their own extent, page-locked memory from elsewhere against the runtime's
 * (hipMemGetAddressRange); anything else -- including a pointer too close to the end of a page-locked block -- takes the copy. */
int ofarn_host_alloc(size_t bytes, void **out);
int ofarn_host_free(void *p);

/* ---- frame front end ------------------------------------------------------
 * cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) on uint8 frames (DenseOF.py:481, 510): OpenCV 4.x fixed point,
 * gray = (3735 B + 19235 G + 9798 R + 16384) >> 15.  h_bgr: n frames of h rows, `stride` bytes per row,
 * 3 bytes per pixel; d_bgr: dense uint8[n][h][w][3].  Gray output dense uint8[n][h][w]. */
int ofarn_bgr2gray(ofarn_ctx *ctx, const uint8_t *h_bgr, int n, int w, int h, int stride, uint8_t *h_gray);
int ofarn_bgr2gray_device(ofarn_ctx *ctx, const uint8_t *d_bgr, int n, int w, int h, uint8_t *d_gray,
                          void *hip_stream);
/* ofarn_calc_batch_device on BGR video frames uint8[n_frames][h][w][3]: the conversion runs on the
 * device per wave in front of the flow (replaces DenseOF.py:510 + :520 for a stack of decoded frames). */
int ofarn_calc_batch_device_bgr(ofarn_ctx *ctx, const uint8_t *d_bgr, int n_frames, int w, int h,
                                int pairs_mode, float *d_flow, uint8_t *d_mask, uint8_t *d_v,
                                void *hip_stream);

/* ---- dense visualisers ----------------------------------------------------
 * draw_hsv (DenseOF.py:109-124): H = uint8((arctan2(fy, fx) + pi) * (180/pi/2)), S = 255,
 * V = uint8(min(4 |f|, 255)), then cv2.cvtColor(hsv, COLOR_HSV2BGR).  flow float32[n][h][w][2];
 * hsv and bgr uint8[n][h][w][3], either may be NULL. */
int ofarn_flow_hsv(ofarn_ctx *ctx, const float *h_flow, int n, int w, int h, uint8_t *h_hsv, uint8_t *h_bgr);
int ofarn_flow_hsv_device(ofarn_ctx *ctx, const float *d_flow, int n, int w, int h, uint8_t *d_hsv,
                          uint8_t *d_bgr, void *hip_stream);
/* cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) alone, uint8[npx][3] -> uint8[npx][3] (DenseOF.py:121). */
int ofarn_hsv2bgr(ofarn_ctx *ctx, const uint8_t *h_hsv, size_t npx, uint8_t *h_bgr);
/* draw_flow (DenseOF.py:40-49): the flow sampled on np.mgrid[step/2:h:step, step/2:w:step] and the
 * int32 line end points [[x, y], [x - fx, y - fy]] (+0.5, truncated) that cv2.polylines receives.
 * ofarn_flow_arrow_count returns K = ny * nx (rows of y, x fastest); lines int32[n][K][2][2]. */
int ofarn_flow_arrow_count(int w, int h, int step, int *nx, int *ny);
int ofarn_flow_arrows(ofarn_ctx *ctx, const float *h_flow, int n, int w, int h, int step, int32_t *h_lines);
int ofarn_flow_arrows_device(ofarn_ctx *ctx, const float *d_flow, int n, int w, int h, int step,
                             int32_t *d_lines, void *hip_stream);

/* ---- sparse pyramidal Lucas-Kanade -----------------------------------------
 * cv2.calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, winSize, maxLevel, criteria, flags,
 * minEigThreshold) as pathfinder_viewer.py:153-158, DenseOF.py:181-185 and SparseOF.py:35-36 call it
 * (OpenCV 4.10 lkpyramid.cpp: buildOpticalFlowPyramid, calcScharrDeriv, LKTrackerInvoker).  The tracker's
 * float sums are taken per window column, then over the columns (oracle OFO_LK_SUM_COLUMNS). */
#define OFARN_LK_USE_INITIAL_FLOW 4    /* cv2.OPTFLOW_USE_INITIAL_FLOW: next_pts holds the initial guesses */
#define OFARN_LK_GET_MIN_EIGENVALS 8   /* cv2.OPTFLOW_LK_GET_MIN_EIGENVALS: err = minimum eigenvalue measure */
typedef struct ofarn_lk_params {
    int win_w, win_h;         /* winSize, each > 2; up to 64 x 255                  default 21 x 21 */
    int max_level;            /* pyramid levels above level 0                       default 3       */
    int max_count;            /* criteria COUNT (clamped to 0..100)                 default 30      */
    double epsilon;           /* criteria EPS (clamped to 0..10)                    default 0.01    */
    int flags;                /* OR of OFARN_LK_*                                   default 0       */
    double min_eig_threshold; /*                                                    default 1e-4    */
} ofarn_lk_params;
void ofarn_lk_default_params(ofarn_lk_params *p);
/* Number of pyramid levels above level 0 that buildOpticalFlowPyramid keeps for this size (<= max_level). */
int ofarn_lk_levels(const ofarn_lk_params *p, int w, int h);
/* One pair, host memory; synchronous.  h_pts, h_next_pts float32[npts][2]; h_status uint8[npts]; h_err
 * float32[npts].  Replaces cv2.calcOpticalFlowPyrLK(prev, next, pts, None, ...). */
int ofarn_lk_calc(ofarn_ctx *ctx, const uint8_t *h_prev, const uint8_t *h_next, int w, int h, int stride,
                  const float *h_pts, int npts, const ofarn_lk_params *p, float *h_next_pts,
                  uint8_t *h_status, float *h_err);
/* A batch of frames on the device (pairs as in ofarn_calc_batch_device).  d_pts: float32[npts][2] shared
 * by all pairs (pts_per_pair = 0) or float32[n_pairs][npts][2]; reverse != 0 tracks from the later frame
 * of a pair to the earlier one, as pathfinder_viewer.py:156 does.  Outputs [n_pairs][npts](x2). */
int ofarn_lk_calc_batch_device(ofarn_ctx *ctx, const uint8_t *d_frames, int n_frames, int w, int h,
                               int pairs_mode, int reverse, const float *d_pts, int npts, int pts_per_pair,
                               const ofarn_lk_params *p, float *d_next_pts, uint8_t *d_status,
                               float *d_err, void *hip_stream);
/* The vector filter + V of pathfinder_viewer.py:159-176, 204-217 on vectors GIVEN at the context's grid
 * points (flow_ = next_pts - points_ of get_flow_lk): vecs float32[n][P][2]. */
int ofarn_vector_filter(ofarn_ctx *ctx, const float *h_vecs, int n, int w, int h, uint8_t *h_mask,
                        uint8_t *h_v, int32_t *h_iflow);
int ofarn_vector_filter_device(ofarn_ctx *ctx, const float *d_vecs, int n, int w, int h, uint8_t *d_mask,
                               uint8_t *d_v, int32_t *d_iflow, void *hip_stream);
/* ---- multi-GPU in one process -------------------------------------------------------
 * The reference processes one pair per loop turn with no state beyond prev_gray (DenseOF.py:519-525), so pairs are independent
 * units: device g of G owns the contiguous range ofarn_shard_pairs(n_pairs, g, G) and there is no data-path collective.  One
 * ofarn_ctx, one host thread and one HIP stream per device; the only exchange is ONE all-gather of the per-pair danger maps
 * (uint8[n_pairs][P] mask and V: 64 x 2304 x 2 B = 295 KB per device for BASELINE config 4) over RCCL, set up with
 * ncclCommInitAll over the listed devices (librccl is dlopen'ed here, not linked).  Flow fields are not gathered. */
typedef struct ofarn_multi ofarn_multi;
/* Contiguous, balanced shard: the first n_pairs % world ranks get one pair more.  Pure arithmetic (no GPU). */
int ofarn_shard_pairs(int n_pairs, int rank, int world, int *start, int *count);
/* Every index of a multi-device batch, pure arithmetic (no GPU; this is what ofarn_multi_calc_batch* computes with).  Arrays of
 * `world` entries (any may be NULL): start / count = the shards; gather_off[r] = byte offset of rank r's rows in the padded all-gather
 * array uint8[world][cap][P] (every rank contributes cap = max count rows; the gather is in place, send = receive + gather_off[r]);
 * global_off[r] = byte offset of rank r's rows in the compact uint8[n_pairs][P] array.  *even: every shard is full, the padded
 * array is the compact one and nothing is moved; otherwise count[r] * P bytes go from gather_off[r] to global_off[r]. */
int ofarn_gather_plan(int n_pairs, int world, int P, int *start, int *count, int *cap, int *even, uint64_t *gather_off,
                      uint64_t *global_off);
/* One persistent host thread per device is created here and joined by ofarn_multi_destroy; all device work of the ofarn_multi_*
 * entry points runs on those threads, so the calling thread's current HIP device is never changed. */
/* devices: n_devices distinct HIP device ordinals (NULL = 0 .. n_devices-1); each gets a context for waves of up to
 * max_batch_per_device pairs. */
int ofarn_multi_create(const ofarn_params *params, const int *devices, int n_devices, int max_w, int max_h,
                       int max_batch_per_device, ofarn_multi **out);
void ofarn_multi_destroy(ofarn_multi *m);
int ofarn_multi_device_count(const ofarn_multi *m);
/* Host frames uint8[n_frames][h][w] (pairs as in ofarn_calc_batch); every device uploads and processes its shard.  h_flow
 * float32[n_pairs][h][w][2] (may be NULL) is filled shard by shard; h_mask / h_v uint8[n_pairs][P] (together, or both NULL) are
 * the all-gathered maps as device 0 holds them after the collective.  Synchronous.  Replaces: the loop DenseOF.py:491-525 over a
 * batch of frames + the filter pathfinder_viewer.py:159-176, 204-217, on G GPUs. */
int ofarn_multi_calc_batch(ofarn_multi *m, const uint8_t *h_frames, int n_frames, int w, int h, int pairs_mode, float *h_flow,
                           uint8_t *h_mask, uint8_t *h_v);
/* Device-resident shards: d_frames[g] / d_flow[g] (d_flow or d_flow[g] may be NULL) are device g's frames and flow output for ITS
 * pairs (ofarn_shard_pairs(n_pairs, g, G)); d_mask_all[g] / d_v_all[g]: uint8[n_pairs][P] on device g, the gathered maps of ALL
 * pairs in global order, on every device.  Enqueued on the devices' internal streams (ofarn_multi_stream), not synchronised. */
int ofarn_multi_calc_batch_device(ofarn_multi *m, const uint8_t *const *d_frames, int n_pairs, int w, int h, int pairs_mode,
                                  float *const *d_flow, uint8_t *const *d_mask_all, uint8_t *const *d_v_all);
int ofarn_multi_synchronize(ofarn_multi *m);
void *ofarn_multi_stream(const ofarn_multi *m, int rank);          /* hipStream_t of a rank */
ofarn_ctx *ofarn_multi_context(const ofarn_multi *m, int rank);    /* its context (profiling, options) */
/* RCCL version code (ncclGetVersion), ncclAllGather calls issued so far, device ms of the last host-variant call on rank 0 */
int ofarn_multi_info(const ofarn_multi *m, int *rccl_version, unsigned long long *allgather_calls, double *last_device_ms);

/* Measurement grid of pathfinder_viewer.py:255-267.  Returns P (number of points, x-major order);
 * if h_pts != NULL writes float32[P][2] = (x, y). */
int ofarn_grid_points(int w, int h, int step, float *h_pts);

/* Vector filter + danger brightness on existing dense flow (pathfinder_viewer.py:159-176, 204-217).
 * flow: float32[n][h][w][2]; mask, v: uint8[n][P]; v is 0 where mask is 0.  iflow (may be NULL):
 * int32[n][P][2], the integer vectors `next_pts - points_` of pathfinder_viewer.py:169-171,178 for
 * EVERY grid point (the reference keeps the rows where mask is set). */
int ofarn_grid_filter(ofarn_ctx *ctx, const float *h_flow, int n, int w, int h, uint8_t *h_mask,
                      uint8_t *h_v, int32_t *h_iflow);
int ofarn_grid_filter_device(ofarn_ctx *ctx, const float *d_flow, int n, int w, int h,
                             uint8_t *d_mask, uint8_t *d_v, int32_t *d_iflow, void *hip_stream);

/* Level plan for a frame size: writes up to cap entries of (w, h, ksize) and sigma per level,
 * level 0 first; returns the number of scales (levels+1 after cropping).  optflowgf.cpp calc(). */
int ofarn_level_plan(const ofarn_params *params, int w, int h, int cap, int *lw, int *lh, int *ksize,
                     double *sigma);

/* Grows the workspace NOW to what a batch call of n_pairs pairs of w x h frames will need (level plan, row-pass and
 * level-image buffers, the matrix buffer of the unfused path, the second workspace of multi-wave batches), so that the call
 * itself allocates nothing.  Needed before a device entry point is recorded into a HIP graph (hipStreamBeginCapture): a
 * call whose workspace would have to grow during capture fails with OFARN_E_INVALID, and no later call may grow the
 * workspace while a captured graph that uses it is alive ("HIP graphs"). */
int ofarn_reserve(ofarn_ctx *ctx, int w, int h, int n_pairs, int pairs_mode);

/* Per-context switches (diagnostics, A/B runs, tests); the environment variable of the same meaning is read once, by
 * ofarn_create.  name: "tile" (-1 choose by grid size / 0 never / 1 always use the tile iteration kernel; OFARN_TILE),
 * "force_generic" (OFARN_FORCE_GENERIC), "row_ltr" (OFARN_ROW_LTR), "direct_min_frames" (OFARN_DIRECT_MIN_FRAMES),
 * "single_stream" (OFARN_SINGLE_STREAM), "stream_zero_copy" (OFARN_STREAM_ZERO_COPY), "stream_overlap" (streaming turn: level
 * build + polynomial expansion on an internal stream beside the iteration chain; 0 off, 1 the chain waits for an event behind every
 * level's expansion, 2 = default: behind the coarsest level's and then every second one's), "push_blocks" (experiment, default 0 = hipMemcpyAsync: ofarn_stream_submit pushes a flow field to page-locked host memory with a
 * kernel of that many blocks), "debug_fail_wave" (test hook: the
 * (value+1)-th wave from now fails with OFARN_E_NOMEM; -1 = off), "prof_dual" (per-kernel timing
 * without forcing the waves of a batch onto one stream), and
 * "box_order" (OFARN_BOX_ORDER): 0 = default, the box window of FarnebackUpdateFlow_Blur summed with restarted running sums (the
 * throughput kernels; oracle order OFO_BOX_BLOCKED); 1 = summed EXACTLY as optflowgf.cpp sums it -- one double running sum per
 * column-channel down the whole image with float row differences, one double running sum along each row (oracle order
 * OFO_BOX_RUNNING) -- bit-identical to that order, unfused, about 3-4 x slower: the mode to use when the last bits must be
 * OpenCV's (the two orders differ by ~1e-6 px mean, 3e-5 max at 1080p).  No effect with OPTFLOW_FARNEBACK_GAUSSIAN. */
int ofarn_set_option(ofarn_ctx *ctx, const char *name, int value);

/* Device time in milliseconds of the most recent host-pointer call (hipEvent, H2D/D2H excluded). */
double ofarn_last_device_ms(const ofarn_ctx *ctx);

/* Per-kernel timing.  While enabled, every kernel launch is bracketed by a hipEvent pair recorded
 * on the stream the kernel is launched on.  ofarn_profile_read() waits for the pending events,
 * then returns one row per (stage, pyramid level) seen since the last read: number of launches,
 * summed device milliseconds and summed work units (level pixels x frames for stages A/B, level
 * pixels x pairs for C/D/E, grid points x pairs for F).  Returns the number of rows (may exceed
 * cap; only cap are written) and resets the accumulators. */
#define OFARN_STAGE_LEVEL_H 0     /* A: u8->f32 + Gaussian row pass at sampled columns */
#define OFARN_STAGE_LEVEL_V 1     /* A: Gaussian column pass + bilinear resize          */
#define OFARN_STAGE_POLYEXP 2     /* B: FarnebackPolyExp                                */
#define OFARN_STAGE_UPSAMPLE 3    /* E: flow resize x 1/pyr_scale                       */
#define OFARN_STAGE_MATRICES 4    /* C: FarnebackUpdateMatrices                         */
#define OFARN_STAGE_BLUR_SOLVE 5  /* D: FarnebackUpdateFlow_Blur                        */
#define OFARN_STAGE_GRID_FILTER 6 /* F: grid sample + vector filter + V                 */
#define OFARN_STAGE_FLOW_ITER 7   /* (E+)C+D fused: one Farneback iteration, M stays on chip */
#define OFARN_STAGE_BGR2GRAY 8    /* front end: cvtColor(COLOR_BGR2GRAY); units = pixels    */
#define OFARN_STAGE_INIT_FLOW 9   /* USE_INITIAL_FLOW: resize(INTER_AREA) * scale           */
#define OFARN_STAGE_COUNT 10
int ofarn_profile_enable(ofarn_ctx *ctx, int on);
int ofarn_profile_read(ofarn_ctx *ctx, int cap, int *stage, int *level, int *launches, double *ms,
                       double *units);

/* Every entry point that takes a context runs on the context's device and leaves the calling thread's current HIP device as it
 * found it.  Test hook for that (a one-GPU box cannot observe a switch): fake_current >= 0 makes the entry points believe the calling
 * thread was on that ordinal (no hipSetDevice back to it is issued), -1 ends it; *scopes = entry points entered by the calling
 * thread so far, *last_restored = the ordinal the most recent one switched back to (-1: it had nothing to restore). */
int ofarn_debug_device_scope(int fake_current, int *scopes, int *last_restored);

/* 1 if a host buffer [p, p + bytes) would be written by the GPU in place (see ofarn_host_alloc), 0 if results are copied into it. */
int ofarn_debug_mapped_host_range(const void *p, size_t bytes);

/* Bytes of HBM workspace held by the context right now.  The polynomial-expansion and flow buffers are sized for
 * max_batch pairs by ofarn_create; level-image, row-pass and (unfused paths only) matrix buffers are allocated by the
 * first call that needs them, so the figure grows until the first batch has run (123 MB per 1080p pair by default). */
uint64_t ofarn_workspace_bytes(const ofarn_ctx *ctx);

/* Library build string: "ofarn <version> gfx950 ...". */
const char *ofarn_version(void);

/* ---- single-stage entry points (host arrays in/out, one image; used by the parity tests to
 * compare each HIP kernel with the oracle stage by stage).  R and M are exchanged as interleaved
 * float32 [h][w][5] (OpenCV's CV_32FC5 order; the device-internal layouts differ), flow as
 * interleaved float32 [h][w][2]. -------------------------------------- */
int ofarn_stage_level_image(ofarn_ctx *ctx, const uint8_t *h_img, int w, int h, int k, float *h_out);
int ofarn_stage_polyexp(ofarn_ctx *ctx, const float *h_img, int w, int h, float *h_R);
int ofarn_stage_update_matrices(ofarn_ctx *ctx, const float *h_R0, const float *h_R1,
                                const float *h_flow, int w, int h, float *h_M);
int ofarn_stage_blur_solve(ofarn_ctx *ctx, const float *h_M, int w, int h, float *h_flow);
int ofarn_stage_flow_upsample(ofarn_ctx *ctx, const float *h_flow, int sw, int sh, int dw, int dh,
                              float *h_out);
/* cv2.pyrDown on uint8 (out: ((w+1)/2) x ((h+1)/2)) and calcScharrDeriv (out: int16[h][w][2]) of the LK path */
int ofarn_stage_pyrdown(ofarn_ctx *ctx, const uint8_t *h_img, int w, int h, uint8_t *h_out);
int ofarn_stage_scharr(ofarn_ctx *ctx, const uint8_t *h_img, int w, int h, int16_t *h_out);
/* resize(flow, INTER_AREA) * mul, shrinking only (the coarsest-level start of USE_INITIAL_FLOW) */
int ofarn_stage_resize_area(ofarn_ctx *ctx, const float *h_flow, int sw, int sh, int dw, int dh, float mul,
                            float *h_out);

/* ==== EXTRAS, NOT PART OF THE SURVEY 8 CONTRACT ===========================================================================
 * Pixel rendering of the viewers' GUI layers (cv2.polylines / cv2.circle / cv2.add rasters).  SURVEY 2 rows 6, 7, 10 and 12 mark
 * these OUT OF SCOPE (GUI); SURVEY 8(f) rank 3 asks only for the HSV wheel and the arrow SAMPLING above.  They were built in round 3,
 * are tested (tests/test_gpu_parity.py, bit-exact against restated cv2 rasters: "parity unpinned" like the rest) and are FROZEN: no
 * further work goes here.  Nothing in the rows of SURVEY 8 depends on them. */
/* -- layers of a view turn (ofarn_stream_next_view) rendered on the device -- */
/* The viewer's obstacle layer of that turn (draw_sparse_lamps, see ofarn_draw_lamps below) from the danger map the turn left on the
 * device: h_out uint8[h][w][3] BGR.  over_frame != 0: the layer added onto the turn's own BGR frame -- output_bgr =
 * cv2.add(output_bgr, draw_sparse_lamps(...)), pathfinder_viewer.py:299-300 -- which the turn had uploaded anyway (needs bgr != 0
 * and h_mask / h_v in that ofarn_stream_next_view call). */
int ofarn_stream_view_lamps(ofarn_ctx *ctx, int w, int h, int radius, int over_frame, uint8_t *h_out);
/* draw_hsv of that turn's flow (which stayed on the device), BGR uint8[h][w][3]; over_frame != 0: added onto the turn's BGR frame,
 * output_bgr = cv2.add(output_bgr, draw_hsv(flow)) (DenseOF.py:577-578). */
int ofarn_stream_view_rainbow(ofarn_ctx *ctx, int w, int h, int over_frame, uint8_t *h_out);
/* draw_flow's image for that turn's flow (ofarn_draw_flow below): the arrow layer, or with over_frame != 0 the turn's BGR frame with
 * the layer added -- output_bgr = cv2.add(output_bgr, draw_flow(shape, flow)) (DenseOF.py:574). */
int ofarn_stream_view_arrows(ofarn_ctx *ctx, int w, int h, int step, int over_frame, uint8_t *h_out);

/* draw_flow as the reference returns it (DenseOF.py:40-59, pathfinder_viewer.py:51-73): the BGR layer uint8[n][h][w][3] with the
 * lines of ofarn_flow_arrows drawn by cv2.polylines(img, lines, False, (0, 255, 0)) -- thickness 1, LINE_8: drawing.cpp clipLine +
 * LineIterator -- and cv2.circle(img, (x1, y1), 1, (0, 255, 0), -1) at every start point.  base (uint8[n][h][w][3] or NULL):
 * cv2.add(base, layer) instead of the layer (DenseOF.py:574); d_base may equal d_out. */
int ofarn_draw_flow(ofarn_ctx *ctx, const float *h_flow, int n, int w, int h, int step, const uint8_t *h_base, uint8_t *h_out);
int ofarn_draw_flow_device(ofarn_ctx *ctx, const float *d_flow, int n, int w, int h, int step, const uint8_t *d_base, uint8_t *d_out,
                           void *hip_stream);

/* cv2.add on uint8 images, n bytes: out = saturate(a + b) -- how the viewers stack their layers onto the frame
 * (DenseOF.py:574-582, pathfinder_viewer.py:297-300).  out may be a or b. */
int ofarn_add_u8(ofarn_ctx *ctx, const uint8_t *h_a, const uint8_t *h_b, size_t n, uint8_t *h_out);
int ofarn_add_u8_device(ofarn_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, size_t n, uint8_t *d_out, void *hip_stream);

/* draw_sparse_lamps (pathfinder_viewer.py:196-222) for danger maps on the context's measurement grid: a BGR layer uint8[n][h][w][3]
 * that is black except for one filled disc per danger point (mask != 0) -- hsv[y, x] = (0, 255, V), cv2.cvtColor(HSV2BGR) = (0, 0, V),
 * cv2.circle(bgr, (x, y), radius, that colour, thickness=-1) in drawing.cpp's LINE_8 raster; the reference's radius is 6.  mask, v:
 * uint8[n][P] as ofarn_calc_batch / ofarn_grid_filter return them.  base (uint8[n][h][w][3], or NULL): the layer is added onto it
 * with saturation, cv2.add(output_bgr, layer) of pathfinder_viewer.py:299-300.  The grid step must exceed 2 * radius (discs that
 * touch would take the colour under their centre: OFARN_E_UNSUPPORTED); radius 0..31. */
int ofarn_draw_lamps(ofarn_ctx *ctx, const uint8_t *h_mask, const uint8_t *h_v, int n, int w, int h, int radius,
                     const uint8_t *h_base, uint8_t *h_out);
int ofarn_draw_lamps_device(ofarn_ctx *ctx, const uint8_t *d_mask, const uint8_t *d_v, int n, int w, int h, int radius,
                            const uint8_t *d_base, uint8_t *d_out, void *hip_stream);

/* The frame layer get_flow_lk returns (pathfinder_viewer.py:147, 180-192): BGR uint8[n][h][w][3], black but for the kept vectors --
 * cv2.polylines(layer, lines, False, (0, 0, 255)) from each kept grid point to point + iflow, then cv2.circle(layer, point, 1,
 * (255, 0, 255), 1) -- and, with draw_bad != 0 (the viewer's key 4), the rejected ones after them in (255, 255, 0).  iflow, mask:
 * what ofarn_vector_filter / ofarn_grid_filter return (iflow is defined at every grid point).  Stack it onto the frame with ofarn_add_u8. */
int ofarn_draw_vectors(ofarn_ctx *ctx, const int32_t *h_iflow, const uint8_t *h_mask, int n, int w, int h, int draw_bad, uint8_t *h_out);
int ofarn_draw_vectors_device(ofarn_ctx *ctx, const int32_t *d_iflow, const uint8_t *d_mask, int n, int w, int h, int draw_bad,
                              uint8_t *d_out, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* OFARN_H */
