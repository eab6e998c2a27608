// Frame-0 initialisation: cv2.goodFeaturesToTrack(roi, 1000, 0.01, 0, blockSize=5,
// useHarrisDetector=True) and cv2.cornerSubPix(im, p, (5,5), (-1,-1), (EPS+MAX_ITER, 100, 0.001)), vidExample.py:110-115.
//
// Harris: Sobel sums and the block x block sums of their products are exact integers (scaled once), the response is a
// fixed float32 expression, the maximum is an order-independent atomic max on an order-preserving integer image of the
// float, local maxima above quality * max are appended as 64-bit keys (response bits << 32 | pixel index) and sorted
// descending with rocPRIM's device radix sort (ties: higher index first, like OpenCV's pointer comparison).
// cornerSubPix: one thread per corner runs OpenCV's iteration verbatim (float32 bilinear patch, float64 accumulation in
// row-major order), so results are bit-identical to the CPU restatement; the Gaussian mask is computed once on the host.
#include <math.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "vh_ws.hpp"

__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u)
{
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

// Sobel 3x3 (aperture 3) with REFLECT_101, packed (dx, dy) int16 per pixel
__global__ __launch_bounds__(256) void k_init_sobel(const uint8_t* im, int w, int h, size_t st, int* dxy)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int xm = vh_reflect101(x - 1, w), xp = vh_reflect101(x + 1, w), ym = vh_reflect101(y - 1, h), yp = vh_reflect101(y + 1, h);
    const uint8_t *r0 = im + (size_t)ym * st, *r1 = im + (size_t)y * st, *r2 = im + (size_t)yp * st;
    const int dx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
    const int dy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
    dxy[(size_t)y * w + x] = (dx & 0xffff) | (dy << 16);
}

// Harris response + global maximum
__global__ __launch_bounds__(256) void k_init_harris(const int* dxy, int w, int h, int block, float s2, float kf, float* resp, unsigned* maxord)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    unsigned o = 0u;
    if (x < w && y < h) {
        const int r0 = block / 2;
        int sxx = 0, sxy = 0, syy = 0;
        for (int j = 0; j < block; j++) {
            const int yy = vh_reflect101(y - r0 + j, h);
            for (int i = 0; i < block; i++) {
                const int xx = vh_reflect101(x - r0 + i, w);
                const int v = dxy[(size_t)yy * w + xx];
                const int a = (int)(short)(v & 0xffff), b = v >> 16;
                sxx += a * a; sxy += a * b; syy += b * b;
            }
        }
        const float a = __fmul_rn((float)sxx, s2), b = __fmul_rn((float)sxy, s2), c = __fmul_rn((float)syy, s2);
        const float tr = __fadd_rn(a, c);
        const float r = __fsub_rn(__fsub_rn(__fmul_rn(a, c), __fmul_rn(b, b)), __fmul_rn(__fmul_rn(kf, tr), tr));
        resp[(size_t)y * w + x] = r;
        o = f2ord(r);
    }
    for (int s = 32; s > 0; s >>= 1) o = max(o, (unsigned)__shfl_xor((int)o, s, 64));
    if ((threadIdx.x & 63) == 0 && o) atomicMax(maxord, o);
}

// local maxima of the thresholded response (interior pixels) -> 64-bit sort keys
__global__ __launch_bounds__(256) void k_init_candidates(const float* resp, int w, int h, const unsigned* maxord, double quality,
                                                         unsigned long long* keys, unsigned* count, unsigned cap)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) return;
    const float thr = (float)((double)ord2f(*maxord) * quality);
    const float v0 = resp[(size_t)y * w + x];
    if (!(v0 > thr)) return;  // THRESH_TOZERO; survivors are compared with the thresholded neighbours
    float m = v0;
#pragma unroll
    for (int j = -1; j <= 1; j++)
#pragma unroll
        for (int i = -1; i <= 1; i++) {
            const float u = resp[(size_t)(y + j) * w + x + i];
            if (u > thr && u > m) m = u;
        }
    if (v0 == m && v0 != 0.f) {
        const unsigned slot = atomicAdd(count, 1u);
        if (slot < cap) keys[slot] = ((unsigned long long)__float_as_uint(v0) << 32) | (unsigned)(y * w + x);
    }
}

__global__ void k_init_emit(const unsigned long long* sorted, const unsigned* count, unsigned cap, int max_corners, int w, float* corners,
                            int* n_out)
{
    const unsigned n = min(min(*count, cap), (unsigned)max_corners);
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_out = (int)n;
    if (i < n) {
        const unsigned idx = (unsigned)(sorted[i] & 0xffffffffull);
        corners[2 * i] = (float)(idx % (unsigned)w);
        corners[2 * i + 1] = (float)(idx / (unsigned)w);
    }
}

// cornerSubPix: one thread per corner
#define SUBPIX_MAXWIN 7
__global__ __launch_bounds__(64) void k_init_subpix(const uint8_t* im, int w, int h, size_t st, float* pts, int n, int win, int max_iter,
                                                    double eps2, const float* mask)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int ww = 2 * win + 1, pw = ww + 2;
    float buf[(2 * SUBPIX_MAXWIN + 3) * (2 * SUBPIX_MAXWIN + 3)];
    const float tx = pts[2 * q], ty = pts[2 * q + 1];
    float cx = tx, cy = ty;
    int iter = 0;
    double err = 0;
    do {
        // getRectSubPix(8u -> 32f), replicated border
        {
            const float ox = __fsub_rn(cx, (float)(pw - 1) * 0.5f), oy = __fsub_rn(cy, (float)(pw - 1) * 0.5f);
            const int ipx = vh_floor(ox), ipy = vh_floor(oy);
            const float a = __fsub_rn(ox, (float)ipx), b = __fsub_rn(oy, (float)ipy);
            const float a11 = __fmul_rn(__fsub_rn(1.f, a), __fsub_rn(1.f, b)), a12 = __fmul_rn(a, __fsub_rn(1.f, b));
            const float a21 = __fmul_rn(__fsub_rn(1.f, a), b), a22 = __fmul_rn(a, b);
            for (int i = 0; i < pw; i++) {
                const int y0 = min(max(ipy + i, 0), h - 1), y1 = min(max(ipy + i + 1, 0), h - 1);
                for (int j = 0; j < pw; j++) {
                    const int x0 = min(max(ipx + j, 0), w - 1), x1 = min(max(ipx + j + 1, 0), w - 1);
                    float v = __fmul_rn((float)im[(size_t)y0 * st + x0], a11);
                    v = __fadd_rn(v, __fmul_rn((float)im[(size_t)y0 * st + x1], a12));
                    v = __fadd_rn(v, __fmul_rn((float)im[(size_t)y1 * st + x0], a21));
                    v = __fadd_rn(v, __fmul_rn((float)im[(size_t)y1 * st + x1], a22));
                    buf[i * pw + j] = v;
                }
            }
        }
        double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
        for (int i = 0; i < ww; i++) {
            const float* sp = buf + (i + 1) * pw + 1;
            const double py = (double)(i - win);
            for (int j = 0; j < ww; j++) {
                const double m = (double)mask[i * ww + j];
                const double tgx = (double)__fsub_rn(sp[j + 1], sp[j - 1]), tgy = (double)__fsub_rn(sp[j + pw], sp[j - pw]);
                const double gxx = __dmul_rn(__dmul_rn(tgx, tgx), m), gxy = __dmul_rn(__dmul_rn(tgx, tgy), m), gyy = __dmul_rn(__dmul_rn(tgy, tgy), m);
                const double px = (double)(j - win);
                a = __dadd_rn(a, gxx); b = __dadd_rn(b, gxy); c = __dadd_rn(c, gyy);
                bb1 = __dadd_rn(bb1, __dadd_rn(__dmul_rn(gxx, px), __dmul_rn(gxy, py)));
                bb2 = __dadd_rn(bb2, __dadd_rn(__dmul_rn(gxy, px), __dmul_rn(gyy, py)));
            }
        }
        const double det = __dsub_rn(__dmul_rn(a, c), __dmul_rn(b, b));
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
        const double sc = __ddiv_rn(1.0, det);
        const float nx = (float)__dsub_rn(__dadd_rn((double)cx, __dmul_rn(__dmul_rn(c, sc), bb1)), __dmul_rn(__dmul_rn(b, sc), bb2));
        const float ny = (float)__dadd_rn(__dsub_rn((double)cy, __dmul_rn(__dmul_rn(b, sc), bb1)), __dmul_rn(__dmul_rn(a, sc), bb2));
        const double ex = (double)__fsub_rn(nx, cx), ey = (double)__fsub_rn(ny, cy);
        err = __dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey));
        cx = nx; cy = ny;
        if (cx < 0 || cx >= (float)w || cy < 0 || cy >= (float)h) break;
    } while (++iter < max_iter && err > eps2);
    if (fabsf(__fsub_rn(cx, tx)) > (float)win || fabsf(__fsub_rn(cy, ty)) > (float)win) { cx = tx; cy = ty; }
    pts[2 * q] = cx; pts[2 * q + 1] = cy;
}

// ---------------------------------------------------------------------------------------------------------------
struct InitScratch {
    int* dxy;
    float* resp;
    unsigned long long *keys, *sorted;
    unsigned* counters;  // [0] max (ordered bits), [1] candidate count
    float* mask;
    void* sort_tmp;
    size_t sort_bytes, pixels;
};
static InitScratch g_init = {};

static int init_reserve(size_t pixels)
{
    if (g_init.pixels >= pixels) return 0;
    (void)hipFree(g_init.dxy); (void)hipFree(g_init.resp); (void)hipFree(g_init.keys); (void)hipFree(g_init.sorted);
    (void)hipFree(g_init.counters); (void)hipFree(g_init.mask); (void)hipFree(g_init.sort_tmp);
    memset(&g_init, 0, sizeof(g_init));
    VH_CHECK(hipMalloc((void**)&g_init.dxy, pixels * 4));
    VH_CHECK(hipMalloc((void**)&g_init.resp, pixels * 4));
    VH_CHECK(hipMalloc((void**)&g_init.keys, pixels * 8));
    VH_CHECK(hipMalloc((void**)&g_init.sorted, pixels * 8));
    VH_CHECK(hipMalloc((void**)&g_init.counters, 16));
    VH_CHECK(hipMalloc((void**)&g_init.mask, sizeof(float) * (2 * SUBPIX_MAXWIN + 1) * (2 * SUBPIX_MAXWIN + 1)));
    size_t bytes = 0;
    VH_CHECK(rocprim::radix_sort_keys_desc(nullptr, bytes, g_init.keys, g_init.sorted, pixels, 0, 64, 0));
    VH_CHECK(hipMalloc(&g_init.sort_tmp, bytes));
    g_init.sort_bytes = bytes;
    g_init.pixels = pixels;
    return 0;
}

extern "C" VH_API int vh_good_features(vh_ctx* c, const uint8_t* im, int w, int h, int stride, int max_corners, double quality, int block,
                                       double k, float* corners, int* count, void* stream)
{
    if (!c || w < 3 || h < 3 || max_corners < 1 || block < 1 || block > 15) return vh_fail(-1, "vh_good_features: bad arguments");
    vh_ctx_bind bound_(c, stream);
    hipStream_t s = bound_.s;
    const size_t pixels = (size_t)w * h;
    int r = init_reserve(pixels);
    if (r) return r;
    const double scale = 1.0 / (4.0 * block * 255.0);
    VH_CHECK(hipMemsetAsync(g_init.counters, 0, 16, s));
    VH_CHECK(hipMemsetAsync(g_init.keys, 0, pixels * 8, s));  // unused tail of the key buffer = 0 keys, which sort last
    dim3 blk(256), grd((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_init_sobel, grd, blk, 0, s, im, w, h, (size_t)stride, g_init.dxy);
    hipLaunchKernelGGL(k_init_harris, grd, blk, 0, s, g_init.dxy, w, h, block, (float)(scale * scale), (float)k, g_init.resp, g_init.counters);
    hipLaunchKernelGGL(k_init_candidates, grd, blk, 0, s, g_init.resp, w, h, g_init.counters, quality, g_init.keys, g_init.counters + 1,
                       (unsigned)pixels);
    // the candidate count stays on the device: the whole (zero padded) key buffer is sorted
    size_t bytes = g_init.sort_bytes;
    VH_CHECK(rocprim::radix_sort_keys_desc(g_init.sort_tmp, bytes, g_init.keys, g_init.sorted, pixels, 0, 64, s));
    hipLaunchKernelGGL(k_init_emit, dim3((max_corners + 255) / 256), dim3(256), 0, s, g_init.sorted, g_init.counters + 1, (unsigned)pixels,
                       max_corners, w, corners, count);
    VH_CHECK(hipGetLastError());
    return 0;
}

extern "C" VH_API int vh_corner_subpix(vh_ctx* c, const uint8_t* im, int w, int h, int stride, float* pts, int n, int win, int max_iter,
                                       double eps, void* stream)
{
    if (!c || win < 1 || win > SUBPIX_MAXWIN || n < 0) return vh_fail(-1, "vh_corner_subpix: bad arguments (window half-size 1..7)");
    if (n == 0) return 0;
    vh_ctx_bind bound_(c, stream);
    hipStream_t s = bound_.s;
    int r = init_reserve(1);
    if (r) return r;
    max_iter = max_iter < 1 ? 1 : (max_iter > 100 ? 100 : max_iter);
    if (eps < 0) eps = 0;
    const int ww = 2 * win + 1;
    float hmask[(2 * SUBPIX_MAXWIN + 1) * (2 * SUBPIX_MAXWIN + 1)];
    for (int i = 0; i < ww; i++) {
        const float y = (float)(i - win) / win, vy = expf(-y * y);
        for (int j = 0; j < ww; j++) {
            const float x = (float)(j - win) / win;
            hmask[i * ww + j] = (float)(vy * expf(-x * x));
        }
    }
    VH_CHECK(hipMemcpyAsync(g_init.mask, hmask, sizeof(float) * ww * ww, hipMemcpyHostToDevice, s));
    VH_CHECK(hipStreamSynchronize(s));  // hmask lives on this stack frame
    hipLaunchKernelGGL(k_init_subpix, dim3((n + 63) / 64), dim3(64), 0, s, im, w, h, (size_t)stride, pts, n, win, max_iter, eps * eps, g_init.mask);
    VH_CHECK(hipGetLastError());
    return 0;
}
