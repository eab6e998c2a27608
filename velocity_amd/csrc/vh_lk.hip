// Pyramidal Lucas-Kanade track solve (K3+K4+K5): one wavefront per track.
//
// Replaces cv2calcOpticalFlowPyrLK (utils/KLT.py:37-51): forward LK over all pyramid levels, optional backward LK from
// the result, forward-backward gate, and the map-back of KLTregional (KLT.py:86-89) / the 1/4-scale stage (KLT.py:115).
// Arithmetic follows SURVEY Appendix A: Scharr derivatives (never materialised: computed on the fly from the 4x4
// neighbourhood of every window sample), 14-bit fixed-point bilinear weights, int16 template / gradient windows kept
// in LDS (lane-private columns, conflict free), exact int64 window sums reduced over the wavefront with a butterfly so
// every lane holds the same sums and the Newton step / stop rules are wave-uniform.
#include "vh_kernels.hpp"

#define W_BITS 14
#define LK_FLT_SCALE (1.f / (1 << 20))

struct Win {
    int w00, w01, w10, w11;
};

__device__ __forceinline__ Win bilinear_weights(float a, float b)
{
    Win w;
    const float ia = __fsub_rn(1.f, a), ib = __fsub_rn(1.f, b);
    w.w00 = vh_round(__fmul_rn(__fmul_rn(ia, ib), (float)(1 << W_BITS)));
    w.w01 = vh_round(__fmul_rn(__fmul_rn(a, ib), (float)(1 << W_BITS)));
    w.w10 = vh_round(__fmul_rn(__fmul_rn(ia, b), (float)(1 << W_BITS)));
    w.w11 = (1 << W_BITS) - w.w00 - w.w01 - w.w10;
    return w;
}

// REFLECT_101 sample (image border of the pyramid levels)
__device__ __forceinline__ int pix_r(const ImgDesc& im, int x, int y)
{
    return im.p[(size_t)vh_reflect101(y, im.h) * im.stride + vh_reflect101(x, im.w)];
}

// template window sample: I (x32), Ix, Iy at the bilinear cell whose top-left pixel is (gx, gy)
template <bool INTERIOR>
__device__ __forceinline__ void template_sample(const ImgDesc& im, int gx, int gy, const Win& w, int& iv, int& ix, int& iy)
{
    int q[4][4];
    if (INTERIOR) {
        const uint8_t* p = im.p + (ptrdiff_t)(gy - 1) * im.stride + (gx - 1);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) q[r][c] = p[r * im.stride + c];
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) q[r][c] = pix_r(im, gx - 1 + c, gy - 1 + r);
    }
    // vertical [3 10 3] smooth and [-1 0 1] difference for the two corner rows, all four columns
    int s[2][4], d[2][4];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            s[b][c] = (q[b][c] + q[b + 2][c]) * 3 + q[b + 1][c] * 10;
            d[b][c] = q[b + 2][c] - q[b][c];
        }
    int gxv[2][2], gyv[2][2];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int a = 0; a < 2; a++) {
            gxv[b][a] = s[b][a + 2] - s[b][a];
            gyv[b][a] = (d[b][a] + d[b][a + 2]) * 3 + d[b][a + 1] * 10;
            if (!INTERIOR) {  // derivative image is constant 0 outside the level
                const bool in = (gx + a) >= 0 && (gx + a) < im.w && (gy + b) >= 0 && (gy + b) < im.h;
                if (!in) { gxv[b][a] = 0; gyv[b][a] = 0; }
            }
        }
    iv = vh_descale(q[1][1] * w.w00 + q[1][2] * w.w01 + q[2][1] * w.w10 + q[2][2] * w.w11, W_BITS - 5);
    ix = vh_descale(gxv[0][0] * w.w00 + gxv[0][1] * w.w01 + gxv[1][0] * w.w10 + gxv[1][1] * w.w11, W_BITS);
    iy = vh_descale(gyv[0][0] * w.w00 + gyv[0][1] * w.w01 + gyv[1][0] * w.w10 + gyv[1][1] * w.w11, W_BITS);
}

template <bool INTERIOR>
__device__ __forceinline__ int search_sample(const ImgDesc& im, int gx, int gy, const Win& w)
{
    int q00, q01, q10, q11;
    if (INTERIOR) {
        const uint8_t* p = im.p + (ptrdiff_t)gy * im.stride + gx;
        q00 = p[0]; q01 = p[1]; q10 = p[im.stride]; q11 = p[im.stride + 1];
    } else {
        q00 = pix_r(im, gx, gy); q01 = pix_r(im, gx + 1, gy); q10 = pix_r(im, gx, gy + 1); q11 = pix_r(im, gx + 1, gy + 1);
    }
    return vh_descale(q00 * w.w00 + q01 * w.w01 + q10 * w.w10 + q11 * w.w11, W_BITS - 5);
}

// One point on one level.  All control flow is wave-uniform.
__device__ void lk_level(const ImgDesc& I, const ImgDesc& J, int win, int level, int top_level, int max_count, double eps2,
                         float p0x, float p0y, float& nxo, float& nyo, int& status, float& err, short* ldsI, int* ldsD, int lane,
                         int& n_iter, int& n_setup)
{
    const float half = (float)(win - 1) * 0.5f;
    const float lscale = (float)(1. / (double)(1 << level));
    float px = __fmul_rn(p0x, lscale), py = __fmul_rn(p0y, lscale);
    float nx, ny;
    if (level == top_level) { nx = px; ny = py; }
    else { nx = __fmul_rn(nxo, 2.f); ny = __fmul_rn(nyo, 2.f); }
    nxo = nx; nyo = ny;

    px = __fsub_rn(px, half); py = __fsub_rn(py, half);
    const int ipx = vh_floor(px), ipy = vh_floor(py);
    if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
        if (level == 0) { status = 0; err = 0.f; }
        return;
    }
    Win w = bilinear_weights(__fsub_rn(px, (float)ipx), __fsub_rn(py, (float)ipy));
    const int npx = win * win;
    const int xinc = 64 % win, yinc = 64 / win;
    const int x_first = lane % win, y_first = lane / win;

    n_setup++;
    long long sA11 = 0, sA12 = 0, sA22 = 0;
    {
        const bool interior = ipx >= 1 && ipy >= 1 && ipx + win + 1 <= I.w - 1 && ipy + win + 1 <= I.h - 1;
        int x = x_first, y = y_first, k = 0;
        for (int i = lane; i < npx; i += 64, k++) {
            int iv, ix, iy;
            if (interior) template_sample<true>(I, ipx + x, ipy + y, w, iv, ix, iy);
            else template_sample<false>(I, ipx + x, ipy + y, w, iv, ix, iy);
            ldsI[k * 64 + lane] = (short)iv;
            ldsD[k * 64 + lane] = (ix & 0xffff) | (iy << 16);
            sA11 += (long long)(ix * ix);
            sA12 += (long long)(ix * iy);
            sA22 += (long long)(iy * iy);
            x += xinc; y += yinc;
            if (x >= win) { x -= win; y++; }
        }
    }
    sA11 = vh_wave_sum_i64(sA11);
    sA12 = vh_wave_sum_i64(sA12);
    sA22 = vh_wave_sum_i64(sA22);
    const float A11 = __fmul_rn((float)sA11, LK_FLT_SCALE), A12 = __fmul_rn((float)sA12, LK_FLT_SCALE), A22 = __fmul_rn((float)sA22, LK_FLT_SCALE);
    float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
    const float dA = __fsub_rn(A11, A22);
    const float disc = __fadd_rn(__fmul_rn(dA, dA), __fmul_rn(__fmul_rn(4.f, A12), A12));
    const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), vh_sqrtf(disc)), (float)(2 * win * win));
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
        if (level == 0) status = 0;
        return;
    }
    D = __fdiv_rn(1.f, D);

    nx = __fsub_rn(nx, half); ny = __fsub_rn(ny, half);
    float pdx = 0.f, pdy = 0.f;
    for (int j = 0; j < max_count; j++) {
        const int inx = vh_floor(nx), iny = vh_floor(ny);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        w = bilinear_weights(__fsub_rn(nx, (float)inx), __fsub_rn(ny, (float)iny));
        const bool interior = inx >= 0 && iny >= 0 && inx + win <= J.w - 1 && iny + win <= J.h - 1;
        n_iter++;
        long long sb1 = 0, sb2 = 0;
        int x = x_first, y = y_first, k = 0;
        for (int i = lane; i < npx; i += 64, k++) {
            const int jv = interior ? search_sample<true>(J, inx + x, iny + y, w) : search_sample<false>(J, inx + x, iny + y, w);
            const int diff = jv - (int)ldsI[k * 64 + lane];
            const int dd = ldsD[k * 64 + lane];
            sb1 += (long long)(diff * (int)(short)(dd & 0xffff));
            sb2 += (long long)(diff * (dd >> 16));
            x += xinc; y += yinc;
            if (x >= win) { x -= win; y++; }
        }
        sb1 = vh_wave_sum_i64(sb1);
        sb2 = vh_wave_sum_i64(sb2);
        const float b1 = __fmul_rn((float)sb1, LK_FLT_SCALE), b2 = __fmul_rn((float)sb2, LK_FLT_SCALE);
        const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, b2), __fmul_rn(A22, b1)), D);
        const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, b1), __fmul_rn(A11, b2)), D);
        nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
        nxo = __fadd_rn(nx, half); nyo = __fadd_rn(ny, half);
        if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= eps2) break;
        if (j > 0 && fabsf(__fadd_rn(dx, pdx)) < 0.01f && fabsf(__fadd_rn(dy, pdy)) < 0.01f) {
            nxo = __fsub_rn(nxo, __fmul_rn(dx, 0.5f));
            nyo = __fsub_rn(nyo, __fmul_rn(dy, 0.5f));
            break;
        }
        pdx = dx; pdy = dy;
    }

    if (status && level == 0) {
        const float fx = __fsub_rn(nxo, half), fy = __fsub_rn(nyo, half);
        const int inx = vh_floor(fx), iny = vh_floor(fy);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { status = 0; return; }
        w = bilinear_weights(__fsub_rn(fx, (float)inx), __fsub_rn(fy, (float)iny));
        const bool interior = inx >= 0 && iny >= 0 && inx + win <= J.w - 1 && iny + win <= J.h - 1;
        long long se = 0;
        int x = x_first, y = y_first, k = 0;
        for (int i = lane; i < npx; i += 64, k++) {
            const int jv = interior ? search_sample<true>(J, inx + x, iny + y, w) : search_sample<false>(J, inx + x, iny + y, w);
            const int diff = jv - (int)ldsI[k * 64 + lane];
            se += (long long)(diff < 0 ? -diff : diff);
            x += xinc; y += yinc;
            if (x >= win) { x -= win; y++; }
        }
        se = vh_wave_sum_i64(se);
        err = __fmul_rn((float)se, __fdiv_rn(1.f, (float)(32 * win * win)));
    }
}

__device__ void lk_track(const PyrDesc& PI, const PyrDesc& PJ, int win, int max_count, double eps2, float px, float py, float& ox,
                         float& oy, int& status, float& err, short* ldsI, int* ldsD, int lane, int& n_iter, int& n_setup)
{
    const int nl = min(PI.nlevels, PJ.nlevels);
    status = 1;
    err = 0.f;
    ox = 0.f; oy = 0.f;
    for (int level = nl - 1; level >= 0; level--)
        lk_level(PI.lv[level], PJ.lv[level], win, level, nl - 1, max_count, eps2, px, py, ox, oy, status, err, ldsI, ldsD, lane, n_iter, n_setup);
}

// grid = (max points, batch), block = one wavefront
__global__ __launch_bounds__(64) void k_lk(const void* job_tab, size_t tab_stride)
{
    const LKJob& job = *reinterpret_cast<const LKJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blockIdx.y * tab_stride);
    const int n = job.n_ptr ? *job.n_ptr : job.n;
    const int pt = blockIdx.x;
    if (pt >= n) return;
    const int lane = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kmax = (job.win * job.win + 63) / 64;
    int* ldsD = reinterpret_cast<int*>(smem);
    short* ldsI = reinterpret_cast<short*>(smem + (size_t)kmax * 64 * 4);

    const float qx = job.p_in[2 * pt], qy = job.p_in[2 * pt + 1];
    const float px = __fsub_rn(__fmul_rn(qx, job.in_scale), job.in_off[0]);
    const float py = __fsub_rn(__fmul_rn(qy, job.in_scale), job.in_off[1]);

    float fx, fy, err;
    int st, n_iter = 0, n_setup = 0;
    lk_track(job.I, job.J, job.win, job.max_count, job.eps2, px, py, fx, fy, st, err, ldsI, ldsD, lane, n_iter, n_setup);
    float fbe = 0.f;
    if (job.fbt >= 0.f) {
        float bx, by, e2;
        int st2;
        lk_track(job.J, job.I, job.win, job.max_count, job.eps2, fx, fy, bx, by, st2, e2, ldsI, ldsD, lane, n_iter, n_setup);
        const float ddx = __fsub_rn(px, bx), ddy = __fsub_rn(py, by);
        fbe = vh_sqrtf(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)));
        st = st && st2 && (fbe < job.fbt);
    }
    if (lane == 0) {
        float ox, oy;
        if (job.out_mode == VH_OUT_SCALE) {
            ox = __fdiv_rn(fx, job.out_scale);
            oy = __fdiv_rn(fy, job.out_scale);
        } else {
            const float ax = __fadd_rn(fx, job.in_off[0]), ay = __fadd_rn(fy, job.in_off[1]);
            if (job.out_mode == VH_OUT_TRANSLATE) {
                ox = __fadd_rn(ax, job.out_off[0]);
                oy = __fadd_rn(ay, job.out_off[1]);
            } else {
                ox = __fadd_rn(__fadd_rn(__fmul_rn(ax, job.T[0]), __fmul_rn(ay, job.T[2])), job.T[4]);
                oy = __fadd_rn(__fadd_rn(__fmul_rn(ax, job.T[1]), __fmul_rn(ay, job.T[3])), job.T[5]);
            }
        }
        job.p_out[2 * pt] = ox;
        job.p_out[2 * pt + 1] = oy;
        job.v_out[pt] = (uint8_t)(st != 0);
        if (job.err_out) job.err_out[pt] = err;
        if (job.fbe_out) job.fbe_out[pt] = fbe;
        if (job.praw_out) { job.praw_out[2 * pt] = fx; job.praw_out[2 * pt + 1] = fy; }
        if (job.stats) {
            atomicAdd(&job.stats[0], (unsigned long long)n_iter);
            atomicAdd(&job.stats[1], (unsigned long long)n_setup);
        }
    }
}

int vh_launch_lk(const void* job_tab, size_t tab_stride, int batch, int max_n, int win, hipStream_t s)
{
    if (max_n <= 0) return 0;
    const int kmax = (win * win + 63) / 64;
    const size_t lds = (size_t)kmax * 64 * 6;
    if (lds > 160 * 1024) return -2;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_lk, dim3(max_n, batch), dim3(64), lds, s, job_tab, tab_stride);
    return 0;
}
