// kernels_generic.hip -- parameter-generic HIP kernels for every stage of the Farneback pipeline.
//
// These kernels accept any (winsize, poly_n, pyr_scale, frame size).  They are the always-correct
// path; kernels_fast.hip holds the specialisations the default parameter set dispatches to.
//
// Arithmetic contract: every kernel performs, per output element, the same IEEE-754 operations in
// the same order as oracle/farneback_oracle.c (box order OFO_BOX_BLOCKED), so results can be
// compared bit for bit.  The library is built with -ffp-contract=off: no FMA contraction.
//
// Which OpenCV function each kernel stands for is given per kernel.
#include "ofarn_internal.h"
#include "farneback_device.h"

namespace ofarn {

// ---------------------------------------------------------------------------------------------
// Stage A, pass 1.  convertTo(CV_32F) + the ROW pass of GaussianBlur (BORDER_REFLECT_101),
// evaluated only at the two source columns (sx, sx+1) each level column will interpolate between.
//   s = k[0]*S[0]; s += k[t]*S[t]   (t = 1..ksize-1, left to right)
// One thread per (y, dx); tmp[f][y][dx] = (value at sx, value at min(sx+1, W-1)).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_level_hpass(const uint8_t *__restrict__ frames,
                                                      size_t frame_stride, int W, int H,
                                                      const float *__restrict__ kern, int ksize,
                                                      const int *__restrict__ xofs, int dw,
                                                      float2 *__restrict__ tmp, int symm)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (dx >= dw) return;
    const uint8_t *row = frames + (size_t)blockIdx.z * frame_stride + (size_t)y * W;
    const int r = ksize >> 1;
    const int sx = xofs[dx];
    const int sx1 = sx + 1 < W ? sx + 1 : W - 1;
    float s0, s1;
    if (symm && (ksize == 3 || ksize == 5)) {
        // SymmRowSmallFilter<float, float> (filter.simd.hpp): centre tap first, then the symmetric pairs
        float a[5], b[5];
        for (int t = 0; t < ksize; t++) {
            a[t] = (float)row[reflect101(sx - r + t, W)];
            b[t] = (float)row[reflect101(sx1 - r + t, W)];
        }
        s0 = row_small_symm(a, kern, ksize);
        s1 = row_small_symm(b, kern, ksize);
        tmp[((size_t)blockIdx.z * H + y) * dw + dx] = make_float2(s0, s1);
        return;
    }
    {
        const float f = kern[0];
        s0 = f * (float)row[reflect101(sx - r, W)];
        s1 = f * (float)row[reflect101(sx1 - r, W)];
    }
    for (int t = 1; t < ksize; t++) {
        const float f = kern[t];
        s0 = s0 + f * (float)row[reflect101(sx - r + t, W)];
        s1 = s1 + f * (float)row[reflect101(sx1 - r + t, W)];
    }
    tmp[((size_t)blockIdx.z * H + y) * dw + dx] = make_float2(s0, s1);
}

// ---------------------------------------------------------------------------------------------
// Stage A, pass 2.  The COLUMN pass of GaussianBlur at source rows (sy, sy+1), then
// resize(INTER_LINEAR): horizontal lerp first, vertical lerp second (resize.cpp HResize/VResize).
//   column: s = k[r]*S[0]; s += k[r+t]*(S[+t] + S[-t])   (t = 1..r)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_level_vpass(const float2 *__restrict__ tmp, int H, int dw,
                                                      int dh, const float *__restrict__ kern,
                                                      int ksize, const float *__restrict__ xa,
                                                      const int *__restrict__ yofs,
                                                      const float *__restrict__ ya,
                                                      float *__restrict__ I)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= dw) return;
    const float2 *t = tmp + (size_t)blockIdx.z * H * dw + dx;
    const int r = ksize >> 1;
    const int sy = yofs[dy];
    const int sy1 = sy + 1 < H ? sy + 1 : H - 1;
    const float f0 = kern[r];
    float2 c0 = t[(size_t)sy * dw], c1 = t[(size_t)sy1 * dw];
    float b00 = f0 * c0.x, b01 = f0 * c0.y, b10 = f0 * c1.x, b11 = f0 * c1.y;
    // unrolled by four so that sixteen loads are in flight per round trip: with one pair on the chip this loop is a chain of
    // memory latencies (20 us at the 79-tap level, 9 with the unroll)
#pragma unroll 4
    for (int k = 1; k <= r; k++) {
        const float f = kern[r + k];
        float2 p = t[(size_t)reflect101(sy + k, H) * dw], m = t[(size_t)reflect101(sy - k, H) * dw];
        b00 = b00 + f * (p.x + m.x);
        b01 = b01 + f * (p.y + m.y);
        p = t[(size_t)reflect101(sy1 + k, H) * dw];
        m = t[(size_t)reflect101(sy1 - k, H) * dw];
        b10 = b10 + f * (p.x + m.x);
        b11 = b11 + f * (p.y + m.y);
    }
    const float a1 = xa[dx], a0 = 1.f - a1;
    const float row0 = b00 * a0 + b01 * a1;
    const float row1 = b10 * a0 + b11 * a1;
    const float w1 = ya[dy], w0 = 1.f - w1;
    I[((size_t)blockIdx.z * dh + dy) * dw + dx] = row0 * w0 + row1 * w1;
}

// ---------------------------------------------------------------------------------------------
// Stage B.  FarnebackPolyExp: (2n+1)-tap separable quadratic fit, replicate borders.
// One 64x16 output tile per block; the level-image tile with an n-pixel halo is staged in LDS,
// the float32 vertical pass writes the (r0, r1, r2) line buffers to LDS, the horizontal pass
// accumulates in double exactly as optflowgf.cpp does (products of b2,b3,b5,b6 are float).
// ---------------------------------------------------------------------------------------------
constexpr int PE_TW = 64, PE_TH = 16;

__global__ __launch_bounds__(256) void k_polyexp(const float *__restrict__ I, float *__restrict__ R,
                                                  int w, int h, PolyCoef c)
{
    extern __shared__ float smem[];
    const int n = c.n;
    const int IW = PE_TW + 2 * n, IH = PE_TH + 2 * n;
    float *sI = smem;                 // [IH][IW]
    float *sR0 = sI + IH * IW;        // [PE_TH][IW]
    float *sR1 = sR0 + PE_TH * IW;
    float *sR2 = sR1 + PE_TH * IW;
    const int x0 = blockIdx.x * PE_TW, y0 = blockIdx.y * PE_TH;
    const size_t npx = (size_t)w * h;
    const float *img = I + (size_t)blockIdx.z * npx;
    const int tid = threadIdx.x;

    for (int i = tid; i < IH * IW; i += 256) {
        const int ly = i / IW, lx = i - ly * IW;
        const int gx = clampi(x0 - n + lx, 0, w - 1), gy = clampi(y0 - n + ly, 0, h - 1);
        sI[i] = img[(size_t)gy * w + gx];
    }
    __syncthreads();
    for (int i = tid; i < PE_TH * IW; i += 256) {
        const int ly = i / IW, lx = i - ly * IW;
        const float *col = sI + (ly + n) * IW + lx;
        float r0 = col[0] * c.g[0], r1 = 0.f, r2 = 0.f;
        for (int k = 1; k <= n; k++) {
            const float a = col[-k * IW], b = col[k * IW];
            const float p = a + b;
            r0 = r0 + c.g[k] * p;
            r1 = r1 + c.xg[k] * (b - a);
            r2 = r2 + c.xxg[k] * p;
        }
        sR0[i] = r0; sR1[i] = r1; sR2[i] = r2;
    }
    __syncthreads();
    float *out = R + (size_t)blockIdx.z * r_frame_stride(npx);
    for (int i = tid; i < PE_TH * PE_TW; i += 256) {
        const int ly = i / PE_TW, ox = i - ly * PE_TW;
        const int gx = x0 + ox, gy = y0 + ly;
        if (gx >= w || gy >= h) continue;
        const float *p0 = sR0 + ly * IW + ox + n, *p1 = sR1 + ly * IW + ox + n, *p2 = sR2 + ly * IW + ox + n;
        const float g0 = c.g[0];
        double b1 = p0[0] * g0, b2 = 0, b3 = p1[0] * g0, b4 = 0, b5 = p2[0] * g0, b6 = 0;
        for (int k = 1; k <= n; k++) {
            const float gk = c.g[k], xgk = c.xg[k], xxgk = c.xxg[k];
            const double tg = p0[k] + p0[-k];
            b1 += tg * gk;
            b4 += tg * xxgk;
            b2 += (p0[k] - p0[-k]) * xgk;
            b3 += (p1[k] + p1[-k]) * gk;
            b6 += (p1[k] - p1[-k]) * xgk;
            b5 += (p2[k] + p2[-k]) * gk;
        }
        const float rv[5] = {(float)(b3 * c.ig11), (float)(b2 * c.ig11), (float)(b1 * c.ig03 + b5 * c.ig33),
                             (float)(b1 * c.ig03 + b4 * c.ig33), (float)(b6 * c.ig55)};
        store_r(out, npx, (unsigned)gy * (unsigned)w + (unsigned)gx, rv);
    }
}

// ---------------------------------------------------------------------------------------------
// Stage E.  resize(prevFlow, INTER_LINEAR) then flow *= 1/pyr_scale (2-channel float).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_upsample(const float2 *__restrict__ src, int sw, int sh,
                                                        float2 *__restrict__ dst, int dw, int dh,
                                                        const int *__restrict__ xofs,
                                                        const float *__restrict__ xa,
                                                        const int *__restrict__ yofs,
                                                        const float *__restrict__ ya, float mul)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= dw) return;
    const float2 *s = src + (size_t)blockIdx.z * sw * sh;
    const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
    const int sy = yofs[dy], sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
    const float a1 = xa[dx], a0 = 1.f - a1, b1 = ya[dy], b0 = 1.f - b1;
    const float2 p00 = s[(size_t)sy * sw + sx], p01 = s[(size_t)sy * sw + sx1];
    const float2 p10 = s[(size_t)sy1 * sw + sx], p11 = s[(size_t)sy1 * sw + sx1];
    const float r0x = p00.x * a0 + p01.x * a1, r0y = p00.y * a0 + p01.y * a1;
    const float r1x = p10.x * a0 + p11.x * a1, r1y = p10.y * a0 + p11.y * a1;
    float2 o;
    o.x = (r0x * b0 + r1x * b1) * mul;
    o.y = (r0y * b0 + r1y * b1) * mul;
    dst[((size_t)blockIdx.z * dh + dy) * dw + dx] = o;
}

// ---------------------------------------------------------------------------------------------
// Stage C.  FarnebackUpdateMatrices: bilinear gather of R1 at (x+dx, y+dy), combine with R0,
// 5-pixel border damping, form G11, G12, G22, h1, h2.  All float32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_matrices(const float *__restrict__ R, int fstep,
                                                          const float2 *__restrict__ flow,
                                                          float *__restrict__ M, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const size_t npx = (size_t)w * h;
    const size_t p = blockIdx.z;
    const float *R0, *R1;
    pair_frames(R, fstep, p, npx, R0, R1);
    const float2 d = flow[p * npx + (size_t)y * w + x];
    float m[5];
    update_matrices_px(R0, R1, npx, w, h, x, y, d.x, d.y, m);
    float *out = M + p * 5 * npx + (size_t)y * w + x;
#pragma unroll
    for (int c = 0; c < 5; c++) out[c * npx] = m[c];
}

// ---------------------------------------------------------------------------------------------
// Stage D.  FarnebackUpdateFlow_Blur: (2m+1)x(2m+1) box sum with replicate borders, scaled by
// 1/winsize^2, then the regularised 2x2 solve.  All sums in double.
//   columns: block-restarted running sums (oracle OFO_BOX_BLOCKED): with B = 2m+1, padded rows
//            r' (row r'-m clamped) and aligned blocks of B rows,
//              T_b = v[Bb] + ... + v[Bb+B-1] (left to right), S_b(j) = S_b(j-1) - v[Bb+j-1],
//              colsum(Bb) = T_b,  colsum(Bb+j) = S_b(j) + P_{b+1}(j-1)
//   rows:    the 2m+1 column sums colsum[clamp(x-m+i)] in chunks of three, left to right:
//            ((e0+e1)+e2) + ((e3+e4)+e5) + ...
// One block = TW columns x one aligned block-row of B output rows; one channel at a time:
// phase A, one thread per column (with halo) marches the B outputs reading M straight from
// HBM/L2 (coalesced across columns); phase B sums horizontally out of LDS.
// ---------------------------------------------------------------------------------------------
constexpr int BS_MAXOUT = 8;   // outputs per thread: TW * B <= 256 * BS_MAXOUT

__global__ __launch_bounds__(256) void k_blur_solve(const float *__restrict__ M, float2 *__restrict__ flow,
                                                     int w, int h, int m, int TW, double scale)
{
    extern __shared__ double sV[];   // [B][IW]
    const int B = 2 * m + 1, IW = TW + 2 * m;
    const int x0 = blockIdx.x * TW, yb = blockIdx.y * B;   // yb: first output row = first padded row
    const size_t npx = (size_t)w * h;
    const int tid = threadIdx.x;
    const int nout = TW * B;
    double acc[BS_MAXOUT][5];

    // unrolled over the channels: `acc` is then indexed by constants only and lives in registers.  Indexed by a loop variable it
    // went to scratch memory (336 B per lane), and the runtime keeps a queue's scratch buffer after the stream is destroyed --
    // every context that ever ran this kernel on its own stream left 17.6 MB of device memory behind.
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const float *src = M + ((size_t)blockIdx.z * 5 + c) * npx;
        for (int i = tid; i < IW; i += 256) {
            const float *col = src + clampi(x0 - m + i, 0, w - 1);
            auto v = [&](int t) { return (double)col[(size_t)clampi(t - m, 0, h - 1) * w]; };
            double P = v(yb);
            for (int j = 1; j < B; j++) P = P + v(yb + j);
            sV[i] = P;
            double S = P, Pn = 0;
            for (int j = 1; j < B && yb + j < h; j++) {
                S = S - v(yb + j - 1);
                const double vn = v(yb + B + j - 1);
                Pn = j == 1 ? vn : Pn + vn;
                sV[j * IW + i] = S + Pn;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BS_MAXOUT; q++) {
            const int idx = tid + 256 * q;
            if (idx < nout) {
                const int ly = idx / TW, ox = idx - ly * TW;
                const double *rowv = sV + ly * IW + ox;
                double s = 0;
                for (int i0 = 0; i0 < B; i0 += 3) {   // chunks of three, left to right (oracle OFO_BOX_BLOCKED)
                    double ch = rowv[i0];
                    if (i0 + 1 < B) ch += rowv[i0 + 1];
                    if (i0 + 2 < B) ch += rowv[i0 + 2];
                    s = i0 == 0 ? ch : s + ch;
                }
                acc[q][c] = s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < BS_MAXOUT; q++) {
        const int idx = tid + 256 * q;
        if (idx >= nout) continue;
        const int ly = idx / TW, ox = idx - ly * TW;
        const int gx = x0 + ox, gy = yb + ly;
        if (gx >= w || gy >= h) continue;
        const double g11 = acc[q][0] * scale, g12 = acc[q][1] * scale, g22 = acc[q][2] * scale;
        const double h1 = acc[q][3] * scale, h2 = acc[q][4] * scale;
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        float2 o;
        o.x = (float)((g11 * h2 - g12 * h1) * idet);
        o.y = (float)((g22 * h1 - g12 * h2) * idet);
        flow[(size_t)blockIdx.z * npx + (size_t)gy * w + gx] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Stage D in OpenCV's LITERAL summation order (ofarn_set_option "box_order" = 1; oracle OFO_BOX_RUNNING).  optflowgf.cpp's
// FarnebackUpdateFlow_Blur keeps ONE double running sum per column-channel down the whole image, updated with the FLOAT difference
// of the entering and the leaving row, and ONE double running sum along each row.  A value therefore depends on every row above
// it and every column left of it in that order: no strip or tile can be started in the middle, which is why the throughput kernels
// use restarted sums (OFO_BOX_BLOCKED, within 3e-5 px of this).  Reproduced here as it is, with the parallelism the order leaves:
//   k_vsum_running:       a thread per (column, channel) marches down ALL rows:  V[c][y][x] = vsum after row y's update (double)
//   k_hsum_running_solve: a thread per row marches along ALL columns:           g += V[x+m] - V[x-m-1], solve, store
// 80 B per pixel of extra HBM traffic and two sequential chains: a verification mode (3-4 x slower), not the default.
// ---------------------------------------------------------------------------------------------
// Both chains run through LDS tiles so that every global access is a whole-wave contiguous segment although one kernel walks
// down columns and the other along rows: k_vsum_running leaves its sums TRANSPOSED (VT[c][x][y]: 64 steps of 64 columns are
// collected in a tile and written as 64 runs of 64 consecutive y), k_hsum_running_solve reads them with its lanes on consecutive
// rows (contiguous again) and transposes its 64 x 64 results back before storing flow rows.  The first form of these kernels
// (plain V[c][y][x], a lane per row striding w * 8 bytes) ran at 5 Gpx/s; see profiles/r04_literal_order.txt.
constexpr int RS_T = 64;                  // threads per block (one wave): columns of k_vsum_running, rows of k_hsum_running_solve
constexpr int RS_S = 32;                  // chain steps collected in LDS before they are written out transposed (16.6 KB per block)

__global__ __launch_bounds__(RS_T) void k_vsum_running(const float *__restrict__ M, double *__restrict__ VT, int w, int h, int m)
{
    extern __shared__ unsigned char rs_smem[];
    double (*tile)[RS_T + 1] = reinterpret_cast<double (*)[RS_T + 1]>(rs_smem);                       // [step][column]
    float *ring = reinterpret_cast<float *>(rs_smem + sizeof(double) * RS_S * (RS_T + 1));            // [2m+2][RS_T]: M rows still in the window
    const int RING = 2 * m + 2;
    const int lane = threadIdx.x, x0 = blockIdx.x * RS_T;
    const int xc = min(x0 + lane, w - 1);                       // lanes beyond the frame redo the last column and store nothing
    const size_t npx = (size_t)w * h;
    const size_t plane = ((size_t)blockIdx.z * 5 + blockIdx.y) * npx;
    const float *src = M + plane + xc;
    double *dst = VT + plane;
    auto fetch = [&](int r) { const float v = src[(size_t)r * w]; ring[(r % RING) * RS_T + lane] = v; return v; };
    auto held = [&](int r) { return ring[(r % RING) * RS_T + lane]; };          // a lane only ever reads what it wrote itself
    double vsum = (double)(fetch(0) * (float)(m + 2));                          // vsum[x] = srow0[x] * (m + 2): a float product
    for (int y = 1; y < m; y++) vsum += (double)(y <= h - 1 ? fetch(y) : held(h - 1));
    const int half = lane >> 5, l32 = lane & 31;
    for (int y0 = 0; y0 < h; y0 += RS_S) {
        const int ny = min(RS_S, h - y0);
        for (int t = 0; t < ny; t++) {
            const int y = y0 + t;
            const float s1 = (y + m <= h - 1) ? fetch(y + m) : held(h - 1);     // rows 0 .. m-1 were fetched above, row y+m >= m is new
            const float s0 = held(y - m - 1 > 0 ? y - m - 1 : 0);
            vsum += (double)(s1 - s0);                                          // the row difference is rounded to float first
            tile[t][lane] = vsum;
        }
        __syncthreads();
        // two columns per store instruction: each half-wave writes RS_S consecutive y of one column (256 B)
        for (int j = 0; j < RS_T; j += 2)
            if (x0 + j + half < w && l32 < ny) dst[(size_t)(x0 + j + half) * h + y0 + l32] = tile[l32][j + half];
        __syncthreads();
    }
}

__global__ __launch_bounds__(RS_T) void k_hsum_running_solve(const double *__restrict__ VT, float2 *__restrict__ flow, int w, int h, int m,
                                                            double scale)
{
    __shared__ float2 tile[RS_S][RS_T + 1];                                     // [column step][row]
    const int lane = threadIdx.x, y0 = blockIdx.x * RS_T;
    const int yc = min(y0 + lane, h - 1);
    const size_t npx = (size_t)w * h;
    const double *v[5];
#pragma unroll
    for (int c = 0; c < 5; c++) v[c] = VT + ((size_t)blockIdx.y * 5 + c) * npx + yc;
    auto at = [&](int c, int x) { return v[c][(size_t)(x < 0 ? 0 : (x > w - 1 ? w - 1 : x)) * h]; };   // the replicated borders of vsum[]
    double g[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        g[c] = at(c, 0) * (m + 2);
        for (int x = 1; x < m; x++) g[c] += at(c, x);
    }
    float2 *out = flow + (size_t)blockIdx.y * npx;
    const int ny = min(RS_T, h - y0);
    const int half = lane >> 5, l32 = lane & 31;
    double in[5], lv[5];
#pragma unroll
    for (int c = 0; c < 5; c++) { in[c] = at(c, m); lv[c] = at(c, -m - 1); }
    for (int x0 = 0; x0 < w; x0 += RS_S) {
        const int nx = min(RS_S, w - x0);
        for (int j = 0; j < nx; j++) {
            const int x = x0 + j;
            double in1[5], lv1[5];
#pragma unroll
            for (int c = 0; c < 5; c++) { in1[c] = at(c, x + 1 + m); lv1[c] = at(c, x - m); }           // the next step's values, ahead of this step's chain
#pragma unroll
            for (int c = 0; c < 5; c++) g[c] += in[c] - lv[c];
            const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale, h1 = g[3] * scale, h2 = g[4] * scale;
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            float2 o;
            o.x = (float)((g11 * h2 - g12 * h1) * idet);
            o.y = (float)((g22 * h1 - g12 * h2) * idet);
            tile[j][lane] = o;
#pragma unroll
            for (int c = 0; c < 5; c++) { in[c] = in1[c]; lv[c] = lv1[c]; }
        }
        __syncthreads();
        // two rows per store instruction: each half-wave writes RS_S consecutive x of one row (256 B)
        for (int r = 0; r < ny; r += 2)
            if (r + half < ny && l32 < nx) out[(size_t)(y0 + r + half) * w + x0 + l32] = tile[l32][r + half];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Stage D, OPTFLOW_FARNEBACK_GAUSSIAN.  FarnebackUpdateFlow_GaussianBlur: separable Gaussian window
// (sigma = 0.3*m, m = winsize/2), float32 throughout, replicate borders, then the same solve.
//   column: s0 = M[y][x]*k[0];  s0 += (M[y+i][x] + M[y-i][x]) * k[i]      (i = 1..m)
//   row   : sum = v[x]*k[0];    sum += k[i] * (v[x-i] + v[x+i])           (i = 1..m)
// One 64x16 tile per block, one channel at a time through LDS.
// ---------------------------------------------------------------------------------------------
constexpr int GS_TW = 64, GS_TH = 16;

__global__ __launch_bounds__(256) void k_gauss_solve(const float *__restrict__ M, float2 *__restrict__ flow,
                                                      int w, int h, int m, const float *__restrict__ kern)
{
    extern __shared__ float gsm[];
    const int IW = GS_TW + 2 * m, IH = GS_TH + 2 * m;
    float *sM = gsm;                 // [IH][IW]
    float *sVf = sM + IH * IW;       // [GS_TH][IW]
    float *sK = sVf + GS_TH * IW;    // [m+1]
    const int x0 = blockIdx.x * GS_TW, y0 = blockIdx.y * GS_TH;
    const size_t npx = (size_t)w * h;
    const int tid = threadIdx.x;
    const int ox = tid & 63, oy = tid >> 6;   // rows oy, oy+4, oy+8, oy+12
    float acc[4][5];
    for (int i = tid; i <= m; i += 256) sK[i] = kern[i];
    for (int c = 0; c < 5; c++) {
        const float *src = M + ((size_t)blockIdx.z * 5 + c) * npx;
        for (int i = tid; i < IH * IW; i += 256) {
            const int ly = i / IW, lx = i - ly * IW;
            const int gx = clampi(x0 - m + lx, 0, w - 1), gy = clampi(y0 - m + ly, 0, h - 1);
            sM[i] = src[(size_t)gy * w + gx];
        }
        __syncthreads();
        for (int i = tid; i < GS_TH * IW; i += 256) {
            const int ly = i / IW, lx = i - ly * IW;
            const float *col = sM + (ly + m) * IW + lx;
            float s0 = col[0] * sK[0];
            for (int k = 1; k <= m; k++) s0 = s0 + (col[k * IW] + col[-k * IW]) * sK[k];
            sVf[i] = s0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float *rowv = sVf + (oy + 4 * q) * IW + ox + m;
            float sum = rowv[0] * sK[0];
            for (int k = 1; k <= m; k++) sum = sum + sK[k] * (rowv[-k] + rowv[k]);
            acc[q][c] = sum;
        }
        __syncthreads();
    }
    const int gx = x0 + ox;
    if (gx >= w) return;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int gy = y0 + oy + 4 * q;
        if (gy >= h) continue;
        const double g11 = acc[q][0], g12 = acc[q][1], g22 = acc[q][2], h1 = acc[q][3], h2 = acc[q][4];
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        float2 o;
        o.x = (float)((g11 * h2 - g12 * h1) * idet);
        o.y = (float)((g22 * h1 - g12 * h2) * idet);
        flow[(size_t)blockIdx.z * npx + (size_t)gy * w + gx] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Stage F.  Sample flow[y][x] at the measurement grid (DenseOF.py:44-45), the vector filter of
// pathfinder_viewer.py:159-176 and the V value of pathfinder_viewer.py:204-217.
// One block per pair, any number of grid points P.  np.median / np.percentile(..., 99) need four order
// statistics of the equalised moduli; they are found with a most-significant-digit radix SELECT on the float
// bit patterns (the moduli are >= 0, so unsigned order is float order): four passes of 8 bits, a 256-bin
// histogram per wanted rank in LDS, the moduli recomputed from the flow on every pass -- no P-sized buffer,
// so a dense grid (step 5 at 1080p: 82 944 points) costs nothing but passes.  All threshold arithmetic is
// float32 in NumPy's order (oracle/filter_oracle.c).  A NaN modulus makes median and percentile NaN, as in NumPy.
//
// arctan2 / cos / sin: NumPy's float32 loops are SIMD approximations (measured on this container's build: up to
// 3.2 ulp for arctan2, 1.4 ulp for cos) whose last bits differ between CPUs, so there is no single "NumPy value".
// The contract here is the correctly rounded float32 result: evaluated in double, rounded once
// (oracle/filter_oracle.c does the same with the host libm); see tests/test_gpu_parity.py for the measured
// agreement with this container's NumPy.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float cr_cosf(float a) { return (float)cos((double)a); }
__device__ __forceinline__ float cr_sinf(float a) { return (float)sin((double)a); }

__device__ __forceinline__ float grid_modulus(const float2 d, const int2 p, float hw, float hh)
{
    const float mod = sqrtf(d.x * d.x + d.y * d.y);
    const float ddx = hw - (float)p.x, ddy = hh - (float)p.y;
    const float mm = sqrtf(ddx * ddx + ddy * ddy);
    return __fdiv_rn(mod, 5.0f + sqrtf(mm)) * 30.0f;
}

__global__ __launch_bounds__(1024) void k_grid_filter(const float2 *__restrict__ flow, int w, int h,
                                                       const int2 *__restrict__ pts, int P, int variant,
                                                       uint8_t *__restrict__ mask, uint8_t *__restrict__ v,
                                                       int *__restrict__ iflow, const float2 *__restrict__ vecs)
{
    __shared__ unsigned s_hist[4][256];
    __shared__ unsigned s_prefix[4], s_rank[4];
    __shared__ int s_nan;
    __shared__ float s_thr[2];
    const int tid = threadIdx.x, nt = blockDim.x;
    // vectors at the grid points: sampled from the dense flow (DenseOF.py:44-45), or given (LK: next_pts - points_)
    const float2 *f = flow ? flow + (size_t)blockIdx.x * w * h : nullptr;
    const float2 *vp = vecs ? vecs + (size_t)blockIdx.x * P : nullptr;
    const float hw = (float)(w / 2), hh = (float)(h / 2);
    auto modulus_at = [&](int i) {
        const int2 p = pts[i];
        const float2 d = vp ? vp[i] : f[(size_t)p.y * w + p.x];
        return grid_modulus(d, p, hw, hh);
    };

    // ranks (0-based, ascending) of the order statistics: the two middles of the median, the two neighbours of the
    // 99-percentile's virtual index (numpy _function_base_impl.py: float32 arithmetic for float32 data)
    const float quant = 99.0f / 100.0f;
    const float vi = (float)(P - 1) * quant;
    const float prevf = floorf(vi);
    int pi = (int)prevf, ni = pi + 1;
    if (vi >= (float)(P - 1)) { pi = P - 1; ni = P - 1; }
    if (ni > P - 1) ni = P - 1;
    if (tid < 4) {
        const int r[4] = {(P & 1) ? P / 2 : P / 2 - 1, P / 2, pi, ni};
        s_rank[tid] = (unsigned)r[tid];
        s_prefix[tid] = 0u;
    }
    if (tid == 0) s_nan = 0;
    // Up to GF_CACHE points per thread keep their vector and modulus in registers (the default 2304-point grid with 1024 threads:
    // three): the select passes and the output loop then read nothing from memory again.  The loads of a thread's points are
    // issued together -- one memory round trip instead of one per point and pass, which is what a single pair's launch consists of
    // (72 -> ~15 us at 1080p; a batch does not notice).  Denser grids fall back to recomputing from the flow on every pass.
    constexpr int GF_CACHE = 4;
    const bool cached = P <= GF_CACHE * nt;
    float2 dc[GF_CACHE];
    float mc[GF_CACHE];
    if (cached) {
        int2 pc[GF_CACHE];
#pragma unroll
        for (int j = 0; j < GF_CACHE; j++) { const int i = tid + j * nt; pc[j] = i < P ? pts[i] : make_int2(0, 0); }
#pragma unroll
        for (int j = 0; j < GF_CACHE; j++) {
            const int i = tid + j * nt;
            dc[j] = i < P ? (vp ? vp[i] : f[(size_t)pc[j].y * w + pc[j].x]) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < GF_CACHE; j++) mc[j] = grid_modulus(dc[j], pc[j], hw, hh);
    }
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 4 * 256; i += nt) (&s_hist[0][0])[i] = 0u;
        __syncthreads();
        const unsigned himask = pass == 0 ? 0u : 0xFFFFFFFFu << (shift + 8);
        const unsigned p0 = s_prefix[0], p1 = s_prefix[1], p2 = s_prefix[2], p3 = s_prefix[3];
        if (cached) {
#pragma unroll
            for (int j = 0; j < GF_CACHE; j++) {
                if (tid + j * nt >= P) continue;
                const float m = mc[j];
                if (m != m) { if (pass == 0) s_nan = 1; continue; }
                const unsigned u = __float_as_uint(m);
                const unsigned hi = u & himask, dg = (u >> shift) & 255u;
                if (hi == p0) atomicAdd(&s_hist[0][dg], 1u);
                if (hi == p1) atomicAdd(&s_hist[1][dg], 1u);
                if (hi == p2) atomicAdd(&s_hist[2][dg], 1u);
                if (hi == p3) atomicAdd(&s_hist[3][dg], 1u);
            }
        } else
        for (int i = tid; i < P; i += nt) {
            const float m = modulus_at(i);
            if (m != m) { if (pass == 0) s_nan = 1; continue; }
            const unsigned u = __float_as_uint(m);
            const unsigned hi = u & himask, dg = (u >> shift) & 255u;
            if (hi == p0) atomicAdd(&s_hist[0][dg], 1u);
            if (hi == p1) atomicAdd(&s_hist[1][dg], 1u);
            if (hi == p2) atomicAdd(&s_hist[2][dg], 1u);
            if (hi == p3) atomicAdd(&s_hist[3][dg], 1u);
        }
        __syncthreads();
        // digit of rank r = the first bin d in [0, 254] whose inclusive count exceeds r, else 255.  One wave per wanted rank: lane l sums
        // bins 4l .. 4l+3, a wave-wide inclusive scan of those sums locates the lane, the lane locates the bin (a single thread walking
        // 256 LDS words one after the other took ~12 us per pass -- most of a one-pair launch)
        if (tid < 256) {
            const int k = tid >> 6, l = tid & 63;
            const unsigned r = s_rank[k];
            const unsigned c0 = s_hist[k][4 * l], c1 = s_hist[k][4 * l + 1], c2 = s_hist[k][4 * l + 2], c3 = s_hist[k][4 * l + 3];
            const unsigned own = (c0 + c1) + (c2 + c3);
            unsigned incl = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
                if (l >= o) incl += t;
            }
            const unsigned excl = incl - own;
            // exactly one lane has excl <= r < incl if r is below the total; bins 255 never stops the walk (it is the fall-through)
            int d = -1;
            unsigned before = 0;
            if (excl <= r && r < incl) {
                if (r < excl + c0) { d = 4 * l; before = excl; }
                else if (r < excl + c0 + c1) { d = 4 * l + 1; before = excl + c0; }
                else if (r < excl + c0 + c1 + c2) { d = 4 * l + 2; before = excl + c0 + c1; }
                else { d = 4 * l + 3; before = excl + c0 + c1 + c2; }
                if (d == 255) before = excl + c0 + c1 + c2;      // same value: the walk would have ended at 255 with acc = everything before it
            }
            const unsigned long long found = __ballot(d >= 0);
            if (found == 0ull && l == 63) { d = 255; before = incl - c3; }      // r beyond the total: d = 255, acc = bins 0 .. 254
            if (d >= 0) {
                s_rank[k] = r - before;
                s_prefix[k] |= (unsigned)d << shift;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float m0 = __uint_as_float(s_prefix[0]), m1 = __uint_as_float(s_prefix[1]);
        float med = (P & 1) ? m1 : (m0 + m1) / 2.0f;
        med = med * 1.0f;
        const float t = vi - prevf;
        const float a = __uint_as_float(s_prefix[2]), b = __uint_as_float(s_prefix[3]);
        const float diff = b - a;
        float p99 = (t >= 0.5f) ? b - diff * (1.0f - t) : a + diff * t;
        if (s_nan) { med = __builtin_nanf(""); p99 = med; }
        s_thr[0] = med;
        s_thr[1] = p99;
    }
    __syncthreads();
    const float med = s_thr[0], p99 = s_thr[1];
    for (int i = tid, j = 0; i < P; i += nt, j++) {
        const int2 p = pts[i];
        float2 d;
        float mod;
        if (cached) {
            d = dc[0]; mod = mc[0];
#pragma unroll
            for (int q = 1; q < GF_CACHE; q++) if (j == q) { d = dc[q]; mod = mc[q]; }
        } else {
            d = vp ? vp[i] : f[(size_t)p.y * w + p.x];
            mod = grid_modulus(d, p, hw, hh);
        }
        const float x = (float)p.x, y = (float)p.y;
        // variant 0: pathfinder_viewer.py:173  (median*1.0 < mod) & (mod < P99)
        // variant 1: DenseOF.py:228            mod > median*1.2   (float32 product)
        const bool keep = variant == 1 ? (mod > med * 1.2f) : ((med < mod) && (mod < p99));
        uint8_t val = 0;
        if (keep || iflow) {
            const float ang = cr_atan2f(d.y, d.x);
            const float gx = mod * cr_cosf(ang), gy = mod * cr_sinf(ang);
            const int nx = (int)((x + gx) + 0.5f), ny = (int)((y + gy) + 0.5f);
            const int px = (int)(x + 0.5f), py = (int)(y + 0.5f);
            const int a = nx - px, b = ny - py;
            if (iflow) {
                iflow[((size_t)blockIdx.x * P + i) * 2] = a;
                iflow[((size_t)blockIdx.x * P + i) * 2 + 1] = b;
            }
            if (keep) {
                double vv = 50.0 + sqrt((double)(a * a + b * b)) * 2.0;
                if (vv > 255.0) vv = 255.0;
                val = (uint8_t)vv;
            }
        }
        mask[(size_t)blockIdx.x * P + i] = keep ? 1 : 0;
        v[(size_t)blockIdx.x * P + i] = val;
    }
}

// ------------------------------------------------------------------------------------ launchers
static inline unsigned cdiv(int a, int b) { return (unsigned)((a + b - 1) / b); }

void launch_level_hpass(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H,
                        int nframes, const float *d_kern, int ksize, const int *d_xofs, int dw, float *tmp, int symm)
{
    dim3 grid(cdiv(dw, 256), H, nframes);
    hipLaunchKernelGGL(k_level_hpass, grid, dim3(256), 0, s, frames, frame_stride, W, H, d_kern, ksize,
                       d_xofs, dw, reinterpret_cast<float2 *>(tmp), symm);
}

void launch_level_vpass(hipStream_t s, const float *tmp, int H, int dw, int dh, int nframes,
                        const float *d_kern, int ksize, const float *d_xa, const int *d_yofs,
                        const float *d_ya, float *I)
{
    dim3 grid(cdiv(dw, 256), dh, nframes);
    hipLaunchKernelGGL(k_level_vpass, grid, dim3(256), 0, s, reinterpret_cast<const float2 *>(tmp), H, dw,
                       dh, d_kern, ksize, d_xa, d_yofs, d_ya, I);
}

void launch_polyexp(hipStream_t s, const float *I, float *R, int w, int h, int nframes, const PolyCoef &c)
{
    const int n = c.n;
    const int IW = PE_TW + 2 * n, IH = PE_TH + 2 * n;
    const size_t lds = sizeof(float) * (size_t)(IH * IW + 3 * PE_TH * IW);
    dim3 grid(cdiv(w, PE_TW), cdiv(h, PE_TH), nframes);
    hipLaunchKernelGGL(k_polyexp, grid, dim3(256), lds, s, I, R, w, h, c);
}

void launch_flow_upsample(hipStream_t s, const float *src, int sw, int sh, float *dst, int dw, int dh,
                          int npairs, const int *d_xofs, const float *d_xa, const int *d_yofs,
                          const float *d_ya, float mul)
{
    dim3 grid(cdiv(dw, 256), dh, npairs);
    hipLaunchKernelGGL(k_flow_upsample, grid, dim3(256), 0, s, reinterpret_cast<const float2 *>(src), sw, sh,
                       reinterpret_cast<float2 *>(dst), dw, dh, d_xofs, d_xa, d_yofs, d_ya, mul);
}

void launch_update_matrices(hipStream_t s, const float *R, int fstep, const float *flow, float *M, int w,
                            int h, int npairs)
{
    dim3 grid(cdiv(w, 256), h, npairs);
    hipLaunchKernelGGL(k_update_matrices, grid, dim3(256), 0, s, R, fstep,
                       reinterpret_cast<const float2 *>(flow), M, w, h);
}

int blur_solve_max_winsize() { return 127; }

void launch_blur_solve(hipStream_t s, const float *M, float *flow, int w, int h, int npairs, int winsize)
{
    const int m = winsize / 2, B = 2 * m + 1;
    int TW = 64;
    while (TW > 8 && (TW * B > 256 * BS_MAXOUT || (size_t)B * (TW + 2 * m) * 8 > 150 * 1024)) TW >>= 1;
    const size_t lds = sizeof(double) * (size_t)B * (TW + 2 * m);
    // the attribute belongs to the (function, device) pair: set it on every launch of this unfused path rather than
    // cache a per-process flag that would be wrong on a second GPU
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_blur_solve), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const double scale = 1. / ((double)winsize * winsize);
    dim3 grid(cdiv(w, TW), cdiv(h, B), npairs);
    hipLaunchKernelGGL(k_blur_solve, grid, dim3(256), lds, s, M, reinterpret_cast<float2 *>(flow), w, h, m, TW,
                       scale);
}

// V: npairs * 5 * w * h doubles of scratch
void launch_blur_solve_running(hipStream_t s, const float *M, double *V, float *flow, int w, int h, int npairs, int winsize)
{
    const int m = winsize / 2;
    const size_t lds = sizeof(double) * RS_S * (RS_T + 1) + sizeof(float) * (size_t)(2 * m + 2) * RS_T;      // 16.6 KB + the M ring (4 KB at winsize 15, 33 KB at 127)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_vsum_running), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_vsum_running, dim3(cdiv(w, RS_T), 5, npairs), dim3(RS_T), lds, s, M, V, w, h, m);
    hipLaunchKernelGGL(k_hsum_running_solve, dim3(cdiv(h, RS_T), npairs), dim3(RS_T), 0, s, V, reinterpret_cast<float2 *>(flow), w, h, m,
                       1. / ((double)winsize * winsize));
}

void launch_gauss_solve(hipStream_t s, const float *M, float *flow, int w, int h, int npairs, int winsize,
                        const float *d_kern)
{
    const int m = winsize / 2;
    const int IW = GS_TW + 2 * m, IH = GS_TH + 2 * m;
    const size_t lds = sizeof(float) * (size_t)(IH * IW + GS_TH * IW + m + 1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gauss_solve), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    dim3 grid(cdiv(w, GS_TW), cdiv(h, GS_TH), npairs);
    hipLaunchKernelGGL(k_gauss_solve, grid, dim3(256), lds, s, M, reinterpret_cast<float2 *>(flow), w, h, m, d_kern);
}

void launch_grid_filter(hipStream_t s, const float *flow, int w, int h, int npairs, const int *d_pts, int P,
                        int variant, uint8_t *mask, uint8_t *v, int32_t *iflow, const float *vecs)
{
    if (npairs <= 0 || P <= 0) return;
    // few pairs: the launch is latency bound, so spread a pair's points over 1024 threads (at most 4 cached points each up to P = 4096);
    // many pairs: 256 threads per pair fill the chip either way
    const int nt = (P >= 4096 || (npairs <= 64 && P > 256)) ? 1024 : 256;
    hipLaunchKernelGGL(k_grid_filter, dim3(npairs), dim3(nt), 0, s,
                       reinterpret_cast<const float2 *>(flow), w, h, reinterpret_cast<const int2 *>(d_pts), P,
                       variant, mask, v, iflow, reinterpret_cast<const float2 *>(vecs));
}

}  // namespace ofarn
