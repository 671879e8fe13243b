// kernels_frontend.hip -- the steps either side of the dense-flow call:
//
//   k_bgr2gray     cv2.cvtColor(img, COLOR_BGR2GRAY) on uint8 frames        DenseOF.py:481,510
//   k_resize_area  resize(flow0, INTER_AREA) * scale, the coarsest-level start of
//                  OPTFLOW_USE_INITIAL_FLOW                                 optflowgf.cpp calc()
//   k_flow_hsv     draw_hsv: direction -> hue, length -> value, HSV2BGR     DenseOF.py:109-124
//   k_flow_arrows  draw_flow: step-14 sampling and int32 line end points    DenseOF.py:40-49
//   k_draw_flow    draw_flow as an image: cv2.polylines + cv2.circle raster  DenseOF.py:40-59
//   k_add_u8       cv2.add on uint8 images (layer compositions)            DenseOF.py:574-582
//   k_draw_lamps   draw_sparse_lamps: a filled disc per danger point,        pathfinder_viewer.py:196-222
//                  optionally cv2.add-ed onto the frame                      pathfinder_viewer.py:299-300
//
// All of them are HBM-bound byte/elementwise work; none is on the timed hot path of bench.py.
#include "ofarn_internal.h"

namespace ofarn {

// ---------------------------------------------------------------------------------------------
// BGR -> gray.  color_rgb.simd.hpp RGB2Gray<uchar>: (b*cb + g*cg + r*cr + (1 << (shift-1))) >> shift.
// One thread converts 4 pixels: three aligned 4-byte loads (12 B) in, one 4-byte store out.
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray,
                                                  size_t npx_total, int cb, int cg, int cr, int shift)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;   // group of 4 pixels
    const size_t p0 = q * 4;
    if (p0 >= npx_total) return;
    const int half = 1 << (shift - 1);
    if (p0 + 4 <= npx_total) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(bgr + p0 * 3);
        const uint32_t a = s[0], b = s[1], c = s[2];   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        const int g0 = (int)((a & 255) * cb + ((a >> 8) & 255) * cg + ((a >> 16) & 255) * cr + half) >> shift;
        const int g1 = (int)((a >> 24) * cb + (b & 255) * cg + ((b >> 8) & 255) * cr + half) >> shift;
        const int g2 = (int)(((b >> 16) & 255) * cb + (b >> 24) * cg + (c & 255) * cr + half) >> shift;
        const int g3 = (int)(((c >> 8) & 255) * cb + ((c >> 16) & 255) * cg + (c >> 24) * cr + half) >> shift;
        *reinterpret_cast<uint32_t *>(gray + p0) = (uint32_t)g0 | ((uint32_t)g1 << 8) | ((uint32_t)g2 << 16) | ((uint32_t)g3 << 24);
    } else {
        for (size_t p = p0; p < npx_total; p++)
            gray[p] = (uint8_t)((bgr[p * 3] * cb + bgr[p * 3 + 1] * cg + bgr[p * 3 + 2] * cr + half) >> shift);
    }
}

// any alignment: one pixel per thread
__global__ __launch_bounds__(256) void k_bgr2gray_bytes(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray,
                                                        size_t npx_total, int cb, int cg, int cr, int shift)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npx_total) return;
    gray[p] = (uint8_t)((bgr[p * 3] * cb + bgr[p * 3 + 1] * cg + bgr[p * 3 + 2] * cr + (1 << (shift - 1))) >> shift);
}

void launch_bgr2gray(hipStream_t s, const uint8_t *bgr, uint8_t *gray, size_t npx_total, int cb, int cg, int cr, int shift)
{
    if (npx_total == 0) return;
    if ((((uintptr_t)bgr | (uintptr_t)gray) & 3) == 0) {
        const size_t groups = (npx_total + 3) / 4;
        hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, bgr, gray, npx_total, cb, cg, cr, shift);
    } else
        hipLaunchKernelGGL(k_bgr2gray_bytes, dim3((unsigned)((npx_total + 255) / 256)), dim3(256), 0, s, bgr, gray, npx_total, cb, cg,
                           cr, shift);
}

// ---------------------------------------------------------------------------------------------
// k_push_host (EXPERIMENT, off by default: ofarn_set_option "push_blocks"): device buffer -> page-locked host memory mapped into the
// device's address space, by a grid of chosen size.  The idea: on this platform a 16.6 MB device-to-host hipMemcpyAsync runs as the
// runtime's blit kernel (__amd_rocclr_copyBuffer, ~0.30 ms) whose grid fills the chip, and in a rocprofv3 timeline of the pipelined frame
// loop the next turn's kernels start only when it has drained; PCIe needs ~100 KB in flight, so a few dozen blocks should saturate it and
// leave the CUs to the next turn.  Measured (profiles/r03_streamprof.txt): 0.79 ms per frame with 8, 16, 32, 64 or 256 blocks against
// 0.57 ms with hipMemcpyAsync -- the push itself takes ~0.6 ms whatever its size when the next turn's chain of small kernels runs
// beside it.  Kept for the record; the copy path is the default.  16 B per lane, four loads in flight per lane.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_push_host(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

void launch_push_host(hipStream_t s, const float *src, float *dst_mapped, size_t nfloats, int blocks)
{
    if (nfloats == 0) return;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_push_host, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<float4 *>(dst_mapped), nfloats / 4);
}

// ---------------------------------------------------------------------------------------------
// resize(INTER_AREA) of 2-channel float flow, shrinking, followed by `flow *= scale`.
// One thread per output element (pair, dy, dx, channel), accumulating in OpenCV's order:
//   general (ResizeArea_Invoker): per source row of the cell  buf = 0; buf = buf + S*alpha (x table order);
//                                 first row: sum = beta*buf, later rows: sum += beta*buf
//   fast (ResizeAreaFast_Invoker, integer factors): sum += S0+S1+S2+S3 four cell pixels at a time, * 1/area
// The source is read through L2 (neighbouring threads read neighbouring cells); this runs once per
// pair at the coarsest scale only.
struct AreaTab {
    const int *xstart, *xsi;    // xstart[dw+1] -> entries of xsi / xalpha
    const float *xalpha;
    const int *ystart, *ysi;
    const float *yalpha;
    int fast, iscale_x, iscale_y;
};

__global__ __launch_bounds__(256) void k_resize_area(const float *__restrict__ src, int sw, int sh, float *__restrict__ dst,
                                                     int dw, int dh, int npairs, AreaTab T, float mul)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_pair = (size_t)dw * dh * 2;
    if (i >= per_pair * npairs) return;
    const int p = (int)(i / per_pair);
    const int r = (int)(i % per_pair);
    const int c = r & 1, dx = (r >> 1) % dw, dy = (r >> 1) / dw;
    const float *S0 = src + (size_t)p * sw * sh * 2 + c;
    float v;
    if (T.fast) {
        const int area = T.iscale_x * T.iscale_y;
        const float *S = S0 + ((size_t)dy * T.iscale_y * sw + (size_t)dx * T.iscale_x) * 2;
        float sum = 0.f;
        int k = 0;
        for (; k <= area - 4; k += 4) {
            const float a0 = S[((size_t)(k / T.iscale_x) * sw + (k % T.iscale_x)) * 2];
            const float a1 = S[((size_t)((k + 1) / T.iscale_x) * sw + ((k + 1) % T.iscale_x)) * 2];
            const float a2 = S[((size_t)((k + 2) / T.iscale_x) * sw + ((k + 2) % T.iscale_x)) * 2];
            const float a3 = S[((size_t)((k + 3) / T.iscale_x) * sw + ((k + 3) % T.iscale_x)) * 2];
            sum += a0 + a1 + a2 + a3;
        }
        for (; k < area; k++) sum += S[((size_t)(k / T.iscale_x) * sw + (k % T.iscale_x)) * 2];
        v = sum * (1.f / (float)area);
    } else {
        float sum = 0.f;
        const int x0 = T.xstart[dx], x1 = T.xstart[dx + 1];
        for (int j = T.ystart[dy]; j < T.ystart[dy + 1]; j++) {
            const float *S = S0 + (size_t)T.ysi[j] * sw * 2;
            float buf = 0.f;
            for (int k = x0; k < x1; k++) buf = buf + S[(size_t)T.xsi[k] * 2] * T.xalpha[k];
            const float t = T.yalpha[j] * buf;
            sum = j == T.ystart[dy] ? t : sum + t;
        }
        v = sum;
    }
    dst[i] = v * mul;
}

void launch_resize_area(hipStream_t s, const float *src, int sw, int sh, float *dst, int dw, int dh, int npairs,
                        const AreaTabHost &t, float mul)
{
    const size_t n = (size_t)dw * dh * 2 * npairs;
    if (n == 0) return;
    AreaTab T{t.xstart, t.xsi, t.xalpha, t.ystart, t.ysi, t.yalpha, t.fast, t.iscale_x, t.iscale_y};
    hipLaunchKernelGGL(k_resize_area, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, sw, sh, dst, dw, dh, npairs, T, mul);
}

// ---------------------------------------------------------------------------------------------
// draw_hsv (DenseOF.py:109-124):
//   ang = arctan2(fy, fx) + pi; v = sqrt(fx*fx + fy*fy)               float32 (NumPy 2 promotion)
//   H = uint8(ang * (180/pi/2)); S = 255; V = uint8(minimum(v*4, 255))  truncating stores
//   bgr = cvtColor(hsv, COLOR_HSV2BGR)                                 color_hsv.simd.hpp HSV2RGB_b
// arctan2 is the one transcendental here.  NumPy's float32 arctan2 is a SIMD approximation (up to 3.2 ulp on this
// container's build, CPU dependent), so the contract is the correctly rounded float32 value: evaluated in double and
// rounded once, as oracle/frontend_oracle.c does; the measured agreement with NumPy is in tests/test_gpu_parity.py.
__device__ __forceinline__ uint8_t sat_round_u8(float v)
{
    const float r = rintf(v);   // cvRound: half to even
    return (uint8_t)(r < 0.f ? 0 : r > 255.f ? 255 : (int)r);
}

__device__ __forceinline__ void hsv2bgr_px(uint8_t H, uint8_t S, uint8_t V, uint8_t out[3])
{
    const float hscale = 6.f / 180.f;
    float h = (float)H, s = (float)S * (1.f / 255.f), v = (float)V * (1.f / 255.f);
    float b, g, r;
    if (s == 0.f) {
        b = g = r = v;
    } else {
        h *= hscale;
        h = fmodf(h, 6.f);
        int sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
        // sector_data = {1,3,0}, {1,0,2}, {3,0,1}, {0,2,1}, {0,1,3}, {2,1,0}  (b, g, r)
        switch (sector) {
        case 0: b = t1; g = t3; r = t0; break;
        case 1: b = t1; g = t0; r = t2; break;
        case 2: b = t3; g = t0; r = t1; break;
        case 3: b = t0; g = t2; r = t1; break;
        case 4: b = t0; g = t1; r = t3; break;
        default: b = t2; g = t1; r = t0; break;
        }
    }
    out[0] = sat_round_u8(b * 255.f);
    out[1] = sat_round_u8(g * 255.f);
    out[2] = sat_round_u8(r * 255.f);
}

// base (or nullptr): the BGR image is added onto it with saturation -- cv2.add(output_bgr, draw_hsv(flow)), DenseOF.py:577-578
__global__ __launch_bounds__(256) void k_flow_hsv(const float2 *__restrict__ flow, size_t npx, uint8_t *__restrict__ hsv,
                                                  uint8_t *__restrict__ bgr, const uint8_t *__restrict__ base)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const float2 f = flow[i];
    const float ang = (float)atan2((double)f.y, (double)f.x) + 3.14159274101257324f;   // correctly rounded arctan2 + float32(np.pi)
    const float v = sqrtf(f.x * f.x + f.y * f.y);
    const uint8_t H = (uint8_t)(int)(ang * 28.6478900909423828f);        // float32(180/np.pi/2)
    const uint8_t V = (uint8_t)(int)fminf(v * 4.f, 255.f);
    if (hsv) { hsv[i * 3] = H; hsv[i * 3 + 1] = 255; hsv[i * 3 + 2] = V; }
    if (bgr) {
        uint8_t o[3];
        hsv2bgr_px(H, 255, V, o);
        if (base) {
#pragma unroll
            for (int k = 0; k < 3; k++) { const int t = (int)o[k] + (int)base[i * 3 + k]; o[k] = (uint8_t)(t > 255 ? 255 : t); }
        }
        bgr[i * 3] = o[0]; bgr[i * 3 + 1] = o[1]; bgr[i * 3 + 2] = o[2];
    }
}

void launch_flow_hsv(hipStream_t s, const float *flow, size_t npx, uint8_t *hsv, uint8_t *bgr, const uint8_t *base)
{
    if (npx == 0) return;
    hipLaunchKernelGGL(k_flow_hsv, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(flow), npx, hsv, bgr, base);
}

// cv2.add on uint8 images: saturating byte-wise sum (the layer compositions of DenseOF.py:574-582, pathfinder_viewer.py:297-300).
// 16 bytes per thread where all three pointers are 16-byte aligned, bytes otherwise.
__global__ __launch_bounds__(256) void k_add_u8(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out,
                                                size_t n, int vec)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        const size_t p = i * 16;
        if (p + 16 <= n) {
            const uint4 x = *reinterpret_cast<const uint4 *>(a + p), y = *reinterpret_cast<const uint4 *>(b + p);
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
            uint32_t r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t t = ((xs[k] >> (8 * j)) & 255u) + ((ys[k] >> (8 * j)) & 255u);
                    acc |= (t > 255u ? 255u : t) << (8 * j);
                }
                r[k] = acc;
            }
            *reinterpret_cast<uint4 *>(out + p) = make_uint4(r[0], r[1], r[2], r[3]);
        } else {
            for (size_t q = p; q < n; q++) { const int t = (int)a[q] + (int)b[q]; out[q] = (uint8_t)(t > 255 ? 255 : t); }
        }
    } else if (i < n) {
        const int t = (int)a[i] + (int)b[i];
        out[i] = (uint8_t)(t > 255 ? 255 : t);
    }
}

void launch_add_u8(hipStream_t s, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n)
{
    if (n == 0) return;
    const int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    const size_t threads = vec ? (n + 15) / 16 : n;
    hipLaunchKernelGGL(k_add_u8, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, b, out, n, vec);
}

__global__ __launch_bounds__(256) void k_hsv2bgr(const uint8_t *__restrict__ hsv, size_t npx, uint8_t *__restrict__ bgr)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    uint8_t o[3];
    hsv2bgr_px(hsv[i * 3], hsv[i * 3 + 1], hsv[i * 3 + 2], o);
    bgr[i * 3] = o[0]; bgr[i * 3 + 1] = o[1]; bgr[i * 3 + 2] = o[2];
}

void launch_hsv2bgr(hipStream_t s, const uint8_t *hsv, size_t npx, uint8_t *bgr)
{
    if (npx == 0) return;
    hipLaunchKernelGGL(k_hsv2bgr, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, hsv, npx, bgr);
}

// ---------------------------------------------------------------------------------------------
// draw_flow (DenseOF.py:40-49): y, x = mgrid[step/2:h:step, step/2:w:step].astype(int);
// lines = int32(vstack([x, y, x - fx, y - fy]).T.reshape(-1, 2, 2) + 0.5).  x - fx is int64 - float32
// = float64 in NumPy; int32() truncates toward zero.
__global__ __launch_bounds__(256) void k_flow_arrows(const float2 *__restrict__ flow, int w, int h, int npairs, int nx, int ny,
                                                     double start, double step, int32_t *__restrict__ lines)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)nx * ny;
    if (i >= per * npairs) return;
    const int p = (int)(i / per), r = (int)(i % per);
    const int iy = r / nx, ix = r % nx;                       // mgrid order: rows of y, x fastest
    const int x = (int)(start + ix * step), y = (int)(start + iy * step);
    const float2 f = flow[(size_t)p * w * h + (size_t)y * w + x];
    int32_t *o = lines + i * 4;
    o[0] = (int32_t)((double)x + 0.5);
    o[1] = (int32_t)((double)y + 0.5);
    o[2] = (int32_t)(((double)x - (double)f.x) + 0.5);
    o[3] = (int32_t)(((double)y - (double)f.y) + 0.5);
}

void launch_flow_arrows(hipStream_t s, const float *flow, int w, int h, int npairs, int nx, int ny, double start, double step,
                        int32_t *lines)
{
    const size_t n = (size_t)nx * ny * npairs;
    if (n == 0) return;
    hipLaunchKernelGGL(k_flow_arrows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(flow), w, h, npairs, nx, ny, start, step, lines);
}

// ---------------------------------------------------------------------------------------------
// draw_flow as an image (DenseOF.py:40-59, pathfinder_viewer.py:51-73): cv2.polylines(img, lines, False, (0, 255, 0)) with the
// default thickness 1 / LINE_8 -- drawing.cpp PolyLine -> ThickLine -> Line: clipLine() to the image, then LineIterator (8-connected
// Bresenham, leftToRight) -- and cv2.circle(img, (x1, y1), 1, (0, 255, 0), -1) at every start point.  Everything drawn has one
// colour, so the image is the union of the pixels whatever the drawing order: one thread per arrow sets the G byte of its pixels
// to 255 in an image the host has initialised (zeros: the layer; a copy of the frame: cv2.add(frame, layer), whose B and R do not
// change and whose G saturates to 255).
__device__ __forceinline__ bool cv_clip_line(long long right, long long bottom, long long &x1, long long &y1, long long &x2, long long &y2)
{
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// Line(img, pt1, pt2): clip if an end point lies outside, then LineIterator(pt1, pt2, 8, leftToRight = true); `put(x, y)` per pixel
template <typename Put>
__device__ __forceinline__ void cv_raster_line(int w, int h, int ax, int ay, int bx, int by, Put put)
{
    long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
    if ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
        (unsigned long long)y1 >= (unsigned long long)h || (unsigned long long)y2 >= (unsigned long long)h)
        if (!cv_clip_line(w - 1, h - 1, x1, y1, x2, y2)) return;
    int px = (int)x1, py = (int)y1;
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int delta_x = 1, delta_y = 1;
    if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; }          // leftToRight: start from the other end
    if (dy < 0) { dy = -dy; delta_y = -1; }
    const bool vert = dy > dx;
    if (vert) { const int t = dx; dx = dy; dy = t; }
    // 8-connected: the major axis advances every step, the minor one when err < 0
    int err = dx - (dy + dy);
    const int plus_delta = dx + dx, minus_delta = -(dy + dy);
    const int major_x = vert ? 0 : delta_x, major_y = vert ? delta_y : 0;
    const int minor_x = vert ? delta_x : 0, minor_y = vert ? 0 : delta_y;
    const int count = dx + 1;
    for (int k = 0; k < count; k++) {
        if ((unsigned)px < (unsigned)w && (unsigned)py < (unsigned)h) put(px, py);
        const int mask = err < 0 ? -1 : 0;
        err += minus_delta + (plus_delta & mask);
        px += major_x + (minor_x & mask);
        py += major_y + (minor_y & mask);
    }
}

// cv2.circle(img, (cx, cy), 1, colour, thickness): Circle()'s radius-1 raster is the four neighbours of the centre, plus the centre
// itself when filled (thickness < 0)
template <typename Put>
__device__ __forceinline__ void cv_raster_circle1(int w, int h, int cx, int cy, bool filled, Put put)
{
    const int ox[5] = {-1, 1, 0, 0, 0}, oy[5] = {0, 0, -1, 1, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (k == 4 && !filled) break;
        const int qx = cx + ox[k], qy = cy + oy[k];
        if ((unsigned)qx < (unsigned)w && (unsigned)qy < (unsigned)h) put(qx, qy);
    }
}

__global__ __launch_bounds__(256) void k_draw_flow(const float2 *__restrict__ flow, int w, int h, int npairs, int nx, int ny,
                                                   double start, double step, uint8_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)nx * ny;
    if (i >= per * npairs) return;
    const int p = (int)(i / per), r = (int)(i % per);
    const int iy = r / nx, ix = r % nx;
    const int gx = (int)(start + ix * step), gy = (int)(start + iy * step);
    const float2 f = flow[(size_t)p * w * h + (size_t)gy * w + gx];
    // the int32 `lines` entry of this arrow (k_flow_arrows)
    const int ax = (int32_t)((double)gx + 0.5), ay = (int32_t)((double)gy + 0.5);
    const int bx = (int32_t)(((double)gx - (double)f.x) + 0.5), by = (int32_t)(((double)gy - (double)f.y) + 0.5);
    uint8_t *img = out + (size_t)p * w * h * 3;
    auto put = [&](int x, int y) { img[((size_t)y * w + x) * 3 + 1] = 255; };
    cv_raster_line(w, h, ax, ay, bx, by, put);
    cv_raster_circle1(w, h, ax, ay, true, put);
}

// One drawing pass of get_flow_lk's frame layer (pathfinder_viewer.py:180-192): for every grid point whose mask equals `want`, either
// the line from the point to point + flow (cv2.polylines) or the radius-1 circle outline at the point (cv2.circle, thickness 1), in one
// colour.  The passes of the reference follow one another (kept lines, kept circles, rejected lines, rejected circles); within a pass
// every writer stores the same colour, so the order inside it does not matter.
__global__ __launch_bounds__(256) void k_draw_vectors(const int2 *__restrict__ pts, const int2 *__restrict__ iflow,
                                                      const uint8_t *__restrict__ mask, int P, int n, int w, int h, int want, int circles,
                                                      int cb, int cg, int cr, uint8_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)P * n) return;
    if ((mask[i] != 0) != (want != 0)) return;
    const int p = (int)(i / P);
    const int2 a = pts[i % P], f = iflow[i];
    uint8_t *img = out + (size_t)p * w * h * 3;
    auto put = [&](int x, int y) {
        uint8_t *q = img + ((size_t)y * w + x) * 3;
        q[0] = (uint8_t)cb; q[1] = (uint8_t)cg; q[2] = (uint8_t)cr;
    };
    if (circles) cv_raster_circle1(w, h, a.x, a.y, false, put);
    else cv_raster_line(w, h, a.x, a.y, a.x + f.x, a.y + f.y, put);
}

void launch_draw_flow(hipStream_t s, const float *flow, int w, int h, int npairs, int nx, int ny, double start, double step, uint8_t *out)
{
    const size_t n = (size_t)nx * ny * npairs;
    if (n == 0) return;
    hipLaunchKernelGGL(k_draw_flow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float2 *>(flow), w, h,
                       npairs, nx, ny, start, step, out);
}

void launch_draw_vectors(hipStream_t s, const int *pts, const int32_t *iflow, const uint8_t *mask, int P, int n, int w, int h,
                         int draw_bad, uint8_t *out)
{
    if ((size_t)P * n == 0) return;
    const dim3 grid((unsigned)(((size_t)P * n + 255) / 256));
    const int2 *p2 = reinterpret_cast<const int2 *>(pts), *f2 = reinterpret_cast<const int2 *>(iflow);
    // pathfinder_viewer.py:183-192: red lines, magenta circles; with draw_bad_flow the rejected vectors in (255, 255, 0) after them
    hipLaunchKernelGGL(k_draw_vectors, grid, dim3(256), 0, s, p2, f2, mask, P, n, w, h, 1, 0, 0, 0, 255, out);
    hipLaunchKernelGGL(k_draw_vectors, grid, dim3(256), 0, s, p2, f2, mask, P, n, w, h, 1, 1, 255, 0, 255, out);
    if (draw_bad) {
        hipLaunchKernelGGL(k_draw_vectors, grid, dim3(256), 0, s, p2, f2, mask, P, n, w, h, 0, 0, 255, 255, 0, out);
        hipLaunchKernelGGL(k_draw_vectors, grid, dim3(256), 0, s, p2, f2, mask, P, n, w, h, 0, 1, 255, 255, 0, out);
    }
}

// ---------------------------------------------------------------------------------------------
// draw_sparse_lamps (pathfinder_viewer.py:196-222) on the measurement grid: hsv[y, x] = (0, 255, V) at every danger point,
// cvtColor(HSV2BGR) -> (0, 0, V), then cv2.circle(bgr, (x, y), radius, that colour, thickness=-1).  The discs of two grid points
// never touch (the host checks step > 2 radius), so a pixel belongs to at most one point: the one whose column index is
// floor((x - x0 + radius) / step).  ext[|dy|] is the half-width of the disc's row |dy| as cv2.circle's midpoint loop fills it.
// With `base` the layer is added to it as cv2.add does (saturating; only the red channel can change).
// One thread writes 4 pixels of the flattened batch: 12 bytes = three dwords when both images are dword-aligned.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_draw_lamps(const uint8_t *__restrict__ mask, const uint8_t *__restrict__ v, int P,
                                                    const uint8_t *__restrict__ base, uint8_t *__restrict__ out, int w, int h,
                                                    size_t npx_total, LampGrid g)
{
    const size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= npx_total) return;
    const size_t npx = (size_t)w * h;
    const int cnt = npx_total - p0 < 4 ? (int)(npx_total - p0) : 4;
    uint8_t px[12];
    if (base) {
        if (ALIGNED && cnt == 4) {
            const uint32_t *b = reinterpret_cast<const uint32_t *>(base + p0 * 3);
            const uint32_t b0 = b[0], b1 = b[1], b2 = b[2];
#pragma unroll
            for (int k = 0; k < 4; k++) { px[k] = (uint8_t)(b0 >> (8 * k)); px[4 + k] = (uint8_t)(b1 >> (8 * k)); px[8 + k] = (uint8_t)(b2 >> (8 * k)); }
        } else {
            for (int k = 0; k < 12; k++) px[k] = k < cnt * 3 ? base[p0 * 3 + k] : 0;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 12; k++) px[k] = 0;
    }
    const int img = (int)(p0 / npx);
    const int r0 = (int)(p0 % npx);
    int y = r0 / w, x = r0 % w, im = img;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < cnt) {
            const int ax = x - g.x0 + g.radius, ay = y - g.y0 + g.radius;
            if (ax >= 0 && ay >= 0) {
                const int i = ax / g.step, j = ay / g.step;
                const int dx = ax - i * g.step - g.radius, dy = ay - j * g.step - g.radius;
                const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
                if (i < g.nx && j < g.ny && ady <= g.radius && adx <= (int)g.ext[ady]) {
                    const size_t q = (size_t)im * P + (size_t)i * g.ny + j;       // grid order: x-major (pathfinder_viewer.py:263-266)
                    if (mask[q]) {
                        const int red = (int)px[3 * k + 2] + (int)v[q];
                        px[3 * k + 2] = (uint8_t)(red > 255 ? 255 : red);
                    }
                }
            }
        }
        if (++x == w) { x = 0; if (++y == h) { y = 0; im++; } }
    }
    if (ALIGNED && cnt == 4) {
        uint32_t *o = reinterpret_cast<uint32_t *>(out + p0 * 3);
        uint32_t o0 = 0, o1 = 0, o2 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { o0 |= (uint32_t)px[k] << (8 * k); o1 |= (uint32_t)px[4 + k] << (8 * k); o2 |= (uint32_t)px[8 + k] << (8 * k); }
        o[0] = o0; o[1] = o1; o[2] = o2;
    } else {
        for (int k = 0; k < cnt * 3; k++) out[p0 * 3 + k] = px[k];
    }
}

void launch_draw_lamps(hipStream_t s, const uint8_t *mask, const uint8_t *v, int P, const uint8_t *base, uint8_t *out, int w, int h,
                       int n, const LampGrid &g)
{
    const size_t total = (size_t)n * w * h;
    if (total == 0) return;
    const dim3 grid((unsigned)((total + 1023) / 1024));
    const bool aligned = (((uintptr_t)out | (uintptr_t)base) & 3) == 0;
    if (aligned) hipLaunchKernelGGL(k_draw_lamps<true>, grid, dim3(256), 0, s, mask, v, P, base, out, w, h, total, g);
    else hipLaunchKernelGGL(k_draw_lamps<false>, grid, dim3(256), 0, s, mask, v, P, base, out, w, h, total, g);
}

}  // namespace ofarn
