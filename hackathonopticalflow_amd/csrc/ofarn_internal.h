// ofarn_internal.h -- shared declarations between the HIP kernels and the C-ABI host code.
//
// Data layout in HBM (all dense, row-major, x fastest):
//   frames   uint8  [F][H][W]                full-resolution input frames
//   tmp      float  [F][H][w_k][2]           row-filtered frame at the two source columns each
//                                            level column samples (level build, stage A)
//   I_k      float  [F][h_k][w_k]            level image
//   R_k      per frame 5*h_k*w_k floats      polynomial expansion in the "4+1" layout: channels
//                                            0..3 of pixel o as float4 [h_k][w_k], then channel 4
//                                            as float [h_k][w_k] (optflowgf.cpp order y,x,yy,xx,xy)
//   M_k      float  [P][5][h_k][w_k]         G11, G12, G22, h1, h2, channel-planar
//   flow_k   float2 [P][h_k][w_k]            (dx, dy)
// F = frames in the wave, P = pairs in the wave.  Pair p reads frames (2p, 2p+1) or (p, p+1).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ofarn {

constexpr int kMaxPolyN = 15;
constexpr int kBorder = 5;   // FarnebackUpdateMatrices: BORDER

// floats between consecutive frames of R (5 per pixel, rounded up so every frame's float4 plane is
// 16-byte aligned)
__host__ __device__ inline size_t r_frame_stride(size_t npx) { return (5 * npx + 3) & ~(size_t)3; }

struct PolyCoef {
    float g[kMaxPolyN + 1];
    float xg[kMaxPolyN + 1];
    float xxg[kMaxPolyN + 1];
    double ig11, ig03, ig33, ig55;
    int n;
};

// Stage A: level image = resize(GaussianBlur(float(frame)))
void launch_level_hpass(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H,
                        int nframes, const float *d_kern, int ksize, const int *d_xofs, int dw,
                        float *tmp, int symm);
void launch_level_vpass(hipStream_t s, const float *tmp, int H, int dw, int dh, int nframes,
                        const float *d_kern, int ksize, const float *d_xa, const int *d_yofs,
                        const float *d_ya, float *I);
// Stage B: polynomial expansion
void launch_polyexp(hipStream_t s, const float *I, float *R, int w, int h, int nframes,
                    const PolyCoef &c);
// Stage E: flow upsample (INTER_LINEAR) and scale by 1/pyr_scale
void launch_flow_upsample(hipStream_t s, const float *src, int sw, int sh, float *dst, int dw, int dh,
                          int npairs, const int *d_xofs, const float *d_xa, const int *d_yofs,
                          const float *d_ya, float mul);
// Stage C: update matrices.  R holds frames; pair p uses frames (p*fstep, p*fstep+1).
void launch_update_matrices(hipStream_t s, const float *R, int fstep, const float *flow, float *M,
                            int w, int h, int npairs);
// Stage D: box average + 2x2 solve
void launch_blur_solve(hipStream_t s, const float *M, float *flow, int w, int h, int npairs,
                       int winsize);
int blur_solve_max_winsize();
// Stage D in OpenCV's literal running-sum order (oracle OFO_BOX_RUNNING); V: npairs * 5 * w * h doubles of scratch
void launch_blur_solve_running(hipStream_t s, const float *M, double *V, float *flow, int w, int h, int npairs, int winsize);
// Stage D with the Gaussian window of OPTFLOW_FARNEBACK_GAUSSIAN; d_kern: m+1 taps (host formula of optflowgf.cpp)
void launch_gauss_solve(hipStream_t s, const float *M, float *flow, int w, int h, int npairs, int winsize,
                        const float *d_kern);
// Stages (E+)C+D fused (kernels_fast.hip).  mode 0: zero input flow; 1: input flow is
// upsample(coarse)*mul computed on the fly; 2: input flow read from flow_in.  flow_out != flow_in.
bool flow_iter_supported(int winsize);
void launch_flow_iter(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w,
                      int h, int npairs, int winsize, int mode, const float *coarse, int cw, int ch,
                      const int *d_xofs, const float *d_xa, const int *d_yofs, const float *d_ya, float mul, int tile_mode = -1);
// The same iteration laid out for latency (kernels_tile.hip): chosen by launch_flow_iter when the marching grid would leave
// most of the chip empty (a single pair, coarse levels of a small batch).  Bit-identical results.
bool flow_iter_tile_supported(int winsize);
bool flow_iter_tile_preferred(long marching_blocks, int tile_mode);
void launch_flow_iter_tile(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w, int h,
                           int npairs, int winsize, int mode, const float *coarse, int cw, int ch, const int *d_xofs,
                           const float *d_xa, float mul);
// The same with the Gaussian window of OPTFLOW_FARNEBACK_GAUSSIAN (kernels_gauss.hip); h_kern: host pointer to the m+1 taps.
bool flow_iter_gauss_supported(int winsize);
void launch_flow_iter_gauss(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w, int h,
                            int npairs, int winsize, const float *h_kern, int mode, const float *coarse, int cw, int ch,
                            const int *d_xofs, const float *d_xa, const int *d_yofs, const float *d_ya, float mul);
// Stage B with compile-time radius, marching layout; with src_is_u8 the level-0 3-tap blur (stage A at
// scale 1) is fused in and `src` are the uint8 frames.  blur3 = host pointer to the 3 kernel taps.
bool polyexp_march_supported(int poly_n);
void launch_polyexp_march(hipStream_t s, const void *src, size_t src_stride, int src_is_u8, float *R, int w, int h,
                          int nframes, const PolyCoef &c, const float *blur3);
// Stage A pass 1 with the frame row staged in LDS (levels >= 1).
// symm (all row passes): 1 = a kernel of 3 or 5 taps is applied in SymmRowSmallFilter's order (row_small_symm), 0 = left to right.
void launch_level_hpass_lds(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                            const float *d_kern, int ksize, const int *d_xofs, int dw, float *tmp, int symm);
// Stage A pass 1 for several levels in one launch (kernels_fast.hip): the frame row is read once.
struct HLevel {
    const float *kern;   // device, ksize taps
    const int *xofs;     // device, dw source columns
    float *dst;          // device, tmp of this level: [F][H][dw][2]
    int dw, ksize;
};
struct HLevels {
    HLevel lv[12];
    int n, rmax;
    int symm;            // see launch_level_hpass_lds
};
size_t hpass_multi_lds_bytes(int W, int rmax);
void launch_level_hpass_multi(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                              const HLevels &L);
// Stage A in one kernel, no intermediate in HBM, for levels that are exactly 1/2, 1/4 or 1/8 of the frame
// (kernels_fast.hip).  h_kern: host pointer to the ksize taps.
bool level_direct_supported(const void *frames, int W, int H, int w, int h, int ksize);
void launch_level_direct(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                         const float *h_kern, int ksize, float *I, int w, int h, int symm);
// Stage A row pass straight from the frames for levels whose width is exactly 1/16, 1/32 or 1/64 of the frame's
// (kernels_fast.hip): fills the same tmp layout as the other row passes.  h_kern: host pointer to the ksize taps.
bool level_hdirect_supported(const void *frames, int W, int w, int ksize);
void launch_level_hdirect(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                          const float *h_kern, int ksize, float *tmp, int w);
// Stage F: grid sample + vector filter + danger brightness.  d_pts int2[P] grid coordinates.
void launch_grid_filter(hipStream_t s, const float *flow, int w, int h, int npairs, const int *d_pts,
                        int P, int variant, uint8_t *mask, uint8_t *v, int32_t *iflow, const float *vecs = nullptr);

// ---- front end / back end of the call (kernels_frontend.hip) ----
// cv2.cvtColor(COLOR_BGR2GRAY) on n pixels of packed BGR; coefficients and shift from color_rgb.simd.hpp.
void launch_bgr2gray(hipStream_t s, const uint8_t *bgr, uint8_t *gray, size_t npx_total, int cb, int cg, int cr, int shift);
// device buffer -> page-locked, device-mapped host memory with a grid of `blocks` blocks (nfloats a multiple of 4, both 16-byte aligned)
void launch_push_host(hipStream_t s, const float *src, float *dst_mapped, size_t nfloats, int blocks);
// resize(INTER_AREA) tables (device pointers), see k_resize_area
struct AreaTabHost {
    int *xstart = nullptr, *xsi = nullptr, *ystart = nullptr, *ysi = nullptr;
    float *xalpha = nullptr, *yalpha = nullptr;
    int fast = 0, iscale_x = 1, iscale_y = 1;
};
void launch_resize_area(hipStream_t s, const float *src, int sw, int sh, float *dst, int dw, int dh, int npairs,
                        const AreaTabHost &t, float mul);
// draw_hsv: flow -> HSV (optional) and BGR (optional), uint8 x 3 per pixel
void launch_flow_hsv(hipStream_t s, const float *flow, size_t npx, uint8_t *hsv, uint8_t *bgr, const uint8_t *base = nullptr);
// draw_flow as an image: sets the G byte of every pixel cv2.polylines / cv2.circle would draw; `out` uint8[npairs][h][w][3] initialised by the caller
void launch_draw_flow(hipStream_t s, const float *flow, int w, int h, int npairs, int nx, int ny, double start, double step, uint8_t *out);
// get_flow_lk's frame layer: lines point -> point + iflow and radius-1 circle outlines for the kept (and, draw_bad, the rejected) grid
// points into `out` uint8[n][h][w][3], initialised by the caller; pts int[P][2], iflow int32[n][P][2], mask uint8[n][P]
void launch_draw_vectors(hipStream_t s, const int *pts, const int32_t *iflow, const uint8_t *mask, int P, int n, int w, int h,
                         int draw_bad, uint8_t *out);
// cv2.add on uint8: out = saturate(a + b), n bytes
void launch_add_u8(hipStream_t s, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void launch_hsv2bgr(hipStream_t s, const uint8_t *hsv, size_t npx, uint8_t *bgr);
// draw_flow: int32 [npairs][ny*nx][2][2] line end points
void launch_flow_arrows(hipStream_t s, const float *flow, int w, int h, int npairs, int nx, int ny, double start, double step,
                        int32_t *lines);

// draw_sparse_lamps on the measurement grid: discs of `radius` at the grid points x0 + i step, y0 + j step whose mask is set
// (point index i ny + j); ext[d] = half-width of the disc's row at distance d from the centre (cv2.circle's filled raster)
constexpr int kMaxLampRadius = 31;
struct LampGrid {
    int x0, y0, step, nx, ny, radius;
    uint8_t ext[kMaxLampRadius + 1];
};
void launch_draw_lamps(hipStream_t s, const uint8_t *mask, const uint8_t *v, int P, const uint8_t *base, uint8_t *out, int w, int h,
                       int n, const LampGrid &g);

// ---- sparse pyramidal Lucas-Kanade (kernels_lk.hip) ----
void launch_pyrdown_u8(hipStream_t s, const uint8_t *src, int sw, int sh, uint8_t *dst, int nframes);
void launch_scharr(hipStream_t s, const uint8_t *src, int w, int h, int16_t *dst, int z0, int zstep, int count);
struct LkLevelArgs {
    const uint8_t *img;      // this level's images of the wave: uint8 [F][h][w]
    const int16_t *deriv;    // Scharr derivatives: int16 [F][h][w][2]
    const float *pts;        // float2 points: [npts] shared by all pairs (pts_stride 0) or [pairs][npts] (pts_stride npts)
    float *next_pts;         // float2 [pairs][npts], in/out across levels
    uint8_t *status;         // [pairs][npts]
    float *err;              // [pairs][npts]
    int w, h, npts, pts_stride;
    int fstep, i_off, j_off; // pair p tracks from frame p*fstep + i_off to frame p*fstep + j_off
    int win_w, win_h, level, top_level, flags, max_count;
    float scale, min_eig;    // (float)(1. / (1 << level)), (float)minEigThreshold
    double eps2;             // epsilon^2
};
size_t lk_track_lds_bytes(int win_w, int win_h);
void launch_lk_track(hipStream_t s, const LkLevelArgs &A, int npairs);

}  // namespace ofarn
