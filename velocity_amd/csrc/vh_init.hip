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

__global__ void k_init_emit(const unsigned long long* sorted, const unsigned* count, unsigned cap, int max_corners, int w, float offx, float offy,
                            float* corners, int* n_out)
{
    const unsigned n = min(min(*count, cap), (unsigned)max_corners);
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_out = (int)n;
    if (i < n) {
        const unsigned idx = (unsigned)(sorted[i] & 0xffffffffull);
        // + the ROI's origin (vidExample.py:110-112: `goodFeaturesToTrack(roi, ...) + np.float32([boxb[0], boxb[2]])`: integer-valued float32, exact)
        corners[2 * i] = __fadd_rn((float)(idx % (unsigned)w), offx);
        corners[2 * i + 1] = __fadd_rn((float)(idx / (unsigned)w), offy);
    }
}

// cornerSubPix: one thread per corner
#define SUBPIX_MAXWIN 7
__global__ __launch_bounds__(64) void k_init_subpix(const uint8_t* im, int w, int h, size_t st, float* pts, int n, const int* n_dev, int win,
                                                    int max_iter, double eps2, const float* mask)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (n_dev ? min(*n_dev, n) : n)) return;  // n_dev: the detector's count, still on the device (vh_frame0_init)
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
// The scratch of the detector lives in the vh_ctx (vh_ws.hpp: InitScratch; round 4 kept ONE process-global copy: two contexts on two streams raced
// on its keys / response / counters, a second device reused the first device's allocation).  It is created by the first frame-0 call of a context,
// sized for max(the context's max_w x max_h, the image at hand), together with the Gaussian masks of cornerSubPix for every window half-size
// (uploaded once, synchronously, at creation: the calls themselves are kernel launches only -- no host synchronisation, legal under stream capture
// once the scratch exists; vh_init_reserve() creates it explicitly).  A larger image than the scratch was made for grows it after waiting for the
// context's stream (refused while that stream is capturing).
static void init_scratch_release(InitScratch& I)
{
    (void)hipFree(I.dxy); (void)hipFree(I.resp); (void)hipFree(I.keys); (void)hipFree(I.sorted);
    (void)hipFree(I.counters); (void)hipFree(I.mask); (void)hipFree(I.sort_tmp);
    memset(&I, 0, sizeof(I));
}
void vh_init_scratch_free(vh_ctx* c)
{
    if (c) init_scratch_release(c->init);
}

static int subpix_mask_offset(int win)  // masks of half-sizes 1 .. win-1 come first
{
    int off = 0;
    for (int k = 1; k < win; k++) off += (2 * k + 1) * (2 * k + 1);
    return off;
}

static int init_reserve(vh_ctx* c, size_t pixels, hipStream_t s)
{
    InitScratch& I = c->init;
    if (I.pixels >= pixels) return 0;
    if (I.pixels) {  // growth: kernels queued through this context may still use the old buffers
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return vh_fail(-6, "frame-0 scratch must grow inside a stream capture: call vh_init_reserve(ctx, w, h) before capturing");
        VH_CHECK(hipStreamSynchronize(s));
    }
    const size_t ctx_px = (size_t)c->max_w * (size_t)c->max_h;
    if (pixels < ctx_px && ctx_px <= ((size_t)1 << 26)) pixels = ctx_px;  // any image the context was created for fits: no growth later
    init_scratch_release(I);
    VH_CHECK(hipMalloc((void**)&I.dxy, pixels * 4));
    VH_CHECK(hipMalloc((void**)&I.resp, pixels * 4));
    VH_CHECK(hipMalloc((void**)&I.keys, pixels * 8));
    VH_CHECK(hipMalloc((void**)&I.sorted, pixels * 8));
    VH_CHECK(hipMalloc((void**)&I.counters, 32));
    const int mask_floats = subpix_mask_offset(SUBPIX_MAXWIN + 1);
    VH_CHECK(hipMalloc((void**)&I.mask, sizeof(float) * mask_floats));
    {   // cornerSubPix's Gaussian window for every half-size, computed once on the host in OpenCV's float32 order (bit-identical to the CPU restatement)
        float* hm = new float[mask_floats];
        for (int win = 1; win <= SUBPIX_MAXWIN; win++) {
            float* m = hm + subpix_mask_offset(win);
            const int ww = 2 * win + 1;
            for (int i = 0; i < ww; i++) {
                const float y = (float)(i - win) / win, vy = expf(-y * y);
                for (int j = 0; j < ww; j++) {
                    const float x = (float)(j - win) / win;
                    m[i * ww + j] = (float)(vy * expf(-x * x));
                }
            }
        }
        const hipError_t e = hipMemcpy(I.mask, hm, sizeof(float) * mask_floats, hipMemcpyHostToDevice);  // synchronous: hm dies here
        delete[] hm;
        VH_CHECK(e);
    }
    size_t bytes = 0;
    VH_CHECK(rocprim::radix_sort_keys_desc(nullptr, bytes, I.keys, I.sorted, pixels, 0, 64, 0));
    VH_CHECK(hipMalloc(&I.sort_tmp, bytes));
    I.sort_bytes = bytes;
    I.pixels = pixels;
    return 0;
}

extern "C" VH_API int vh_init_reserve(vh_ctx* c, int w, int h, void* stream)
{
    if (!c || w < 1 || h < 1) return vh_fail(-1, "vh_init_reserve: bad arguments");
    VH_BIND(c, stream);
    return init_reserve(c, (size_t)w * h, bound_.s);
}

// goodFeaturesToTrack on the stream s; (offx, offy) is added to every corner (the ROI origin); count stays on the device
static int good_features_run(vh_ctx* c, const uint8_t* im, int w, int h, int stride, int max_corners, double quality, int block, double k, float offx,
                             float offy, float* corners, int* count, hipStream_t s)
{
    const size_t pixels = (size_t)w * h;
    int r = init_reserve(c, pixels, s);
    if (r) return r;
    InitScratch& I = c->init;
    const double scale = 1.0 / (4.0 * block * 255.0);
    VH_CHECK(hipMemsetAsync(I.counters, 0, 16, s));
    VH_CHECK(hipMemsetAsync(I.keys, 0, pixels * 8, s));  // unused tail of the key buffer = 0 keys, which sort last
    dim3 blk(256), grd((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_init_sobel, grd, blk, 0, s, im, w, h, (size_t)stride, I.dxy);
    hipLaunchKernelGGL(k_init_harris, grd, blk, 0, s, I.dxy, w, h, block, (float)(scale * scale), (float)k, I.resp, I.counters);
    hipLaunchKernelGGL(k_init_candidates, grd, blk, 0, s, I.resp, w, h, I.counters, quality, I.keys, I.counters + 1, (unsigned)pixels);
    // the candidate count stays on the device: the whole (zero padded) key buffer is sorted
    size_t bytes = I.sort_bytes;
    VH_CHECK(rocprim::radix_sort_keys_desc(I.sort_tmp, bytes, I.keys, I.sorted, pixels, 0, 64, s));
    hipLaunchKernelGGL(k_init_emit, dim3((max_corners + 255) / 256), dim3(256), 0, s, I.sorted, I.counters + 1, (unsigned)pixels, max_corners, w, offx,
                       offy, corners, count);
    VH_CHECK(hipGetLastError());
    return 0;
}

extern "C" VH_API int vh_good_features(vh_ctx* c, const uint8_t* im, int w, int h, int stride, int max_corners, double quality, int block,
                                       double k, float* corners, int* count, void* stream)
{
    if (!c || w < 3 || h < 3 || max_corners < 1 || block < 1 || block > 15) return vh_fail(-1, "vh_good_features: bad arguments");
    VH_BIND(c, stream);
    return good_features_run(c, im, w, h, stride, max_corners, quality, block, k, 0.f, 0.f, corners, count, bound_.s);
}

static int corner_subpix_run(vh_ctx* c, const uint8_t* im, int w, int h, int stride, float* pts, int n, const int* n_dev, int win, int max_iter,
                             double eps, hipStream_t s)
{
    int r = init_reserve(c, 1, s);
    if (r) return r;
    max_iter = max_iter < 1 ? 1 : (max_iter > 100 ? 100 : max_iter);
    if (eps < 0) eps = 0;
    hipLaunchKernelGGL(k_init_subpix, dim3((n + 63) / 64), dim3(64), 0, s, im, w, h, (size_t)stride, pts, n, n_dev, win, max_iter, eps * eps,
                       c->init.mask + subpix_mask_offset(win));
    VH_CHECK(hipGetLastError());
    return 0;
}

extern "C" VH_API int vh_corner_subpix(vh_ctx* c, const uint8_t* im, int w, int h, int stride, float* pts, int n, int win, int max_iter,
                                       double eps, void* stream)
{
    if (!c || win < 1 || win > SUBPIX_MAXWIN || n < 0) return vh_fail(-1, "vh_corner_subpix: bad arguments (window half-size 1..7)");
    if (n == 0) return 0;
    VH_BIND(c, stream);
    return corner_subpix_run(c, im, w, h, stride, pts, n, nullptr, win, max_iter, eps, bound_.s);
}

// ---------------------------------------------------------------------------------------------------------------
// vh_frame0_init: vidExample.py:105-127 as one device-resident sequence
// ---------------------------------------------------------------------------------------------------------------
struct Frame0Job {
    double K[9];
    float q[8];          // the clicked plate corners (vidExample.py:104-106)
    double plate[12];    // worldPointsLicensePlate (common.py:150-156), 4 x 3
    int boxa[4];         // boundingRect(q, border 0): x0 x1 y0 y1
    int max_n;           // 4 + max_corners
};

// p[0:4] = q (vidExample.py:116 `p = np.concatenate((q, p))`: the corners found sit behind them already)
__global__ void k_frame0_head(Frame0Job J, float* p, double* plate)
{
    if (threadIdx.x < 8) p[threadIdx.x] = J.q[threadIdx.x];
    if (threadIdx.x < 12) plate[threadIdx.x] = J.plate[threadIdx.x];
}

// p3 = addcol0(image2world(K, R, t, p).astype(float)) @ R + t (vidExample.py:119, common.py:49-55); vp = insidebbox(p, boxa) (:126, images.py:22-27);
// n_out = 4 + corners found.  float64 throughout (the reference's float32 inverse carries ~1e-7; the contract is 1e-4): H = [R[0:2]; t] @ K, its
// inverse by the adjugate, q = [x y 1] @ inv(H), (X, Y) = q[0:2] / q[2], p3 = X R[0] + Y R[1] + t
__global__ __launch_bounds__(256) void k_frame0_world(Frame0Job J, const double* R, const float* t, const int* n_corners, const float* p, double* p3,
                                                      uint8_t* vp, int* n_out)
{
    const int n = min(4 + *n_corners, J.max_n);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_out = n;
    if (i >= J.max_n) return;
    if (i >= n) { vp[i] = 0; p3[3 * i] = 0; p3[3 * i + 1] = 0; p3[3 * i + 2] = 0; return; }
    const double td[3] = {(double)t[0], (double)t[1], (double)t[2]};
    double H[9];
    for (int c = 0; c < 3; c++) {
        H[0 + c] = R[0] * J.K[0 + c] + R[1] * J.K[3 + c] + R[2] * J.K[6 + c];
        H[3 + c] = R[3] * J.K[0 + c] + R[4] * J.K[3 + c] + R[5] * J.K[6 + c];
        H[6 + c] = td[0] * J.K[0 + c] + td[1] * J.K[3 + c] + td[2] * J.K[6 + c];
    }
    const double c00 = H[4] * H[8] - H[5] * H[7], c01 = H[5] * H[6] - H[3] * H[8], c02 = H[3] * H[7] - H[4] * H[6];
    const double det = H[0] * c00 + H[1] * c01 + H[2] * c02;
    const double Hi[9] = {c00 / det, (H[2] * H[7] - H[1] * H[8]) / det, (H[1] * H[5] - H[2] * H[4]) / det,
                          c01 / det, (H[0] * H[8] - H[2] * H[6]) / det, (H[2] * H[3] - H[0] * H[5]) / det,
                          c02 / det, (H[1] * H[6] - H[0] * H[7]) / det, (H[0] * H[4] - H[1] * H[3]) / det};
    const float xf = p[2 * i], yf = p[2 * i + 1];
    const double x = (double)xf, y = (double)yf;
    const double q0 = x * Hi[0] + y * Hi[3] + Hi[6];
    const double q1 = x * Hi[1] + y * Hi[4] + Hi[7];
    const double q2 = x * Hi[2] + y * Hi[5] + Hi[8];
    const double X = q0 / q2, Y = q1 / q2;
    p3[3 * i] = X * R[0] + Y * R[3] + td[0];
    p3[3 * i + 1] = X * R[1] + Y * R[4] + td[1];
    p3[3 * i + 2] = X * R[2] + Y * R[5] + td[2];
    vp[i] = (xf > (float)J.boxa[0] && xf < (float)J.boxa[1] && yf > (float)J.boxa[2] && yf < (float)J.boxa[3]) ? 1 : 0;
}

// boundingRect(x, imshape, border) of a few host points (images.py:9-19; floor on all four edges like the device kernel k_bounding_rect)
static void host_bounding_rect(const float* q, int n, int imw, int imh, int bx, int by, int* roi)
{
    float mnx = q[0], mxx = q[0], mny = q[1], mxy = q[1];
    for (int i = 1; i < n; i++) {
        mnx = fminf(mnx, q[2 * i]); mxx = fmaxf(mxx, q[2 * i]);
        mny = fminf(mny, q[2 * i + 1]); mxy = fmaxf(mxy, q[2 * i + 1]);
    }
    int x0 = (int)floorf(mnx), y0 = (int)floorf(mny);
    const int bw = (int)floorf(mxx) - x0 + 1, bh = (int)floorf(mxy) - y0 + 1;
    int x1 = x0 + bw + bx, y1 = y0 + bh + by;
    x0 -= bx; y0 -= by;
    roi[0] = x0 > 1 ? x0 : 1; roi[1] = x1 < imw ? x1 : imw; roi[2] = y0 > 1 ? y0 : 1; roi[3] = y1 < imh ? y1 : imh;
}

extern "C" VH_API int vh_frame0_init(vh_ctx* c, const uint8_t* im, int w, int h, int stride, const float* q_host, const double* K_host,
                                     const double* plate_host, int border_x, int border_y, int max_corners, double quality, int block, double k,
                                     int subpix_win, int subpix_iter, double subpix_eps, float* p_out, double* p3_out, uint8_t* vp_out, float* t_out,
                                     double* R_out, double* res_out, int* n_out, int* roi_host, void* stream)
{
    if (!c || !im || !q_host || !K_host || !plate_host || !p_out || !p3_out || !vp_out || !t_out || !R_out || !res_out || !n_out)
        return vh_fail(-1, "vh_frame0_init: null argument");
    if (w < 3 || h < 3 || stride < w || max_corners < 1 || block < 1 || block > 15 || subpix_win < 1 || subpix_win > SUBPIX_MAXWIN)
        return vh_fail(-1, "vh_frame0_init: bad arguments");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    Frame0Job J;
    memset(&J, 0, sizeof(J));
    for (int i = 0; i < 9; i++) J.K[i] = K_host[i];
    for (int i = 0; i < 8; i++) J.q[i] = q_host[i];
    for (int i = 0; i < 12; i++) J.plate[i] = plate_host[i];
    J.max_n = 4 + max_corners;
    int boxb[4];
    host_bounding_rect(q_host, 4, w, h, 0, 0, J.boxa);           // vidExample.py:107
    host_bounding_rect(q_host, 4, w, h, border_x, border_y, boxb);  // :108
    if (roi_host) for (int i = 0; i < 4; i++) { roi_host[i] = J.boxa[i]; roi_host[4 + i] = boxb[i]; }
    const int rw = boxb[1] - boxb[0], rh = boxb[3] - boxb[2];
    if (rw < 3 || rh < 3) return vh_fail(-1, "vh_frame0_init: the plate ROI is empty");
    {
        int r0 = init_reserve(c, (size_t)rw * rh, s);
        if (r0) return r0;
    }
    // Harris corners of the ROI view im[boxb[2]:boxb[3], boxb[0]:boxb[1]] (:109-112), refined on the full image (:113-115), behind the 4 plate corners (:116)
    int* n_corners = reinterpret_cast<int*>(c->init.counters) + 4;  // the detector's count (own word: k_frame0_world's blocks read it while one of them writes n_out)
    int r = good_features_run(c, im + (size_t)boxb[2] * stride + boxb[0], rw, rh, stride, max_corners, quality, block, k, (float)boxb[0], (float)boxb[2],
                              p_out + 8, n_corners, s);
    if (r) return r;
    r = corner_subpix_run(c, im, w, h, stride, p_out + 8, max_corners, n_corners, subpix_win, subpix_iter, subpix_eps, s);
    if (r) return r;
    // plate pose from the 4 corners (:118): estimateWorldCameraPose(K, q, plate, findR=True) from x0 = [rpy(I), (0, 0, 1)] (NLS.py:9,20)
    double* plate_dev = c->d_small + 32;
    hipLaunchKernelGGL(k_frame0_head, dim3(1), dim3(64), 0, s, J, p_out, plate_dev);
    PoseJob P;
    memset(&P, 0, sizeof(P));
    for (int i = 0; i < 9; i++) { P.K[i] = K_host[i]; P.R[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    P.x0[5] = 1.0;
    P.p = p_out; P.pw = plate_dev; P.n = 4; P.mode = 1;
    P.t_out = t_out; P.R_out = R_out; P.res_out = res_out; P.p_proj = nullptr; P.info_out = reinterpret_cast<int*>(c->init.counters + 2);
    static_assert(sizeof(PoseJob) <= sizeof(LKJob), "PoseJob must fit in the LKJob slot");
    PoseJob* d = reinterpret_cast<PoseJob*>(&c->d_ws[0].lk);  // parked like every stateless call's descriptor
    VH_CHECK(vh_store(d, P, s));
    vh_launch_pose(d, sizeof(PoseJob), 1, 1, 4, s);
    hipLaunchKernelGGL(k_frame0_world, dim3((J.max_n + 255) / 256), dim3(256), 0, s, J, R_out, t_out, n_corners, p_out, p3_out, vp_out, n_out);
    VH_CHECK(hipGetLastError());
    return 0;
}
