// Pyramidal Lucas-Kanade track solve (K3+K4+K5).
//
// Replaces cv2calcOpticalFlowPyrLK (utils/KLT.py:37-51): forward LK over all pyramid levels, optional backward LK from
// the result, forward-backward gate, and the map-back of KLTregional (KLT.py:86-89) / the 1/4-scale stage (KLT.py:115).
// Arithmetic follows SURVEY Appendix A: Scharr derivatives (never materialised), 14-bit fixed-point bilinear weights,
// int16 template / gradient windows, exact integer window sums converted once to float32, so every implementation in this
// file returns the same bits.  Four implementations, routed by vh_launch_lk (window size and number of tracks in flight):
//   k_lk            per-sample reference kernel, one wavefront per track (any window; fallback for windows > 63 px)
//   k_lk_strip<W>   strips of 4 samples per lane on packed 16-bit dot products, one wavefront per track, loads from L1/L2
//   k_lk3<W,NW,M>   LDS-staged patch + search region, NW wavefronts per track, vertical strip runs (the 51x51 fine stage)
//   k_lk_q<W>       4 tracks per wavefront, one 16-lane DPP row per track, template in registers (the 15x15 coarse stages)
#include <atomic>

#include "vh_kernels.hpp"
#include "vh_valu.hpp"

#define W_BITS 14
#define LK_FLT_SCALE (1.f / (1 << 20))

// (float)v for |v| < 2^53 with ONE rounding: hi * 2^32 + lo is exact in float64, the final conversion rounds once --
// identical to the int64 -> float32 conversion, in 5 instructions instead of the compiler's ~15-instruction sequence
// The per-track inputs of a launch (count, launch order, start points) hang off pointers that were themselves loaded from the job descriptor: read through a
// GLOBAL pointer (where the address is workgroup uniform they then come through the scalar cache), not through the generic one hipcc assumes -- a
// workgroup otherwise starts with a chain of three dependent flat loads before its first image row is requested
typedef const int __attribute__((address_space(1)))* gptr_i32;
typedef const float __attribute__((address_space(1)))* gptr_f32;
#define LK_N_OF(job) ((job).n_ptr ? *(gptr_i32)(job).n_ptr : (job).n)

__device__ __forceinline__ float i64_to_f32(long long v)
{
    const double d = __dadd_rn(__dmul_rn((double)(int)(v >> 32), 4294967296.0), (double)(unsigned)(v & 0xffffffffll));
    return (float)d;
}

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
    return im.p[(size_t)vh_reflect101_near(y, im.h) * im.stride + vh_reflect101_near(x, im.w)];  // (|x|, |y| within a window of the level)
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
__device__ void lk_level(const ImgDesc I, const ImgDesc J, int win, int level, int top_level, int max_count, double eps2,
                         float p0x, float p0y, float& nxo, float& nyo, int& status, float& err, short* ldsI, int* ldsD, int lane,
                         int& n_iter, int& n_setup, bool want_err)
{
    const float half = (float)(win - 1) * 0.5f;
    const float lscale = __uint_as_float((unsigned)(127 - level) << 23);  // 2^-level exactly = (float)(1. / (1 << level)), without the f64 division
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
    const float A11 = __fmul_rn(i64_to_f32(sA11), LK_FLT_SCALE), A12 = __fmul_rn(i64_to_f32(sA12), LK_FLT_SCALE), A22 = __fmul_rn(i64_to_f32(sA22), LK_FLT_SCALE);
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
        const float b1 = __fmul_rn(i64_to_f32(sb1), LK_FLT_SCALE), b2 = __fmul_rn(i64_to_f32(sb2), LK_FLT_SCALE);
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
        if (!want_err) return;  // err is discarded by the caller (KLT.py:83 `pa, v, _`): only the bounds rule matters
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
                         float& oy, int& status, float& err, short* ldsI, int* ldsD, int lane, int& n_iter, int& n_setup, bool want_err)
{
    const int nl = min(PI.nlevels, PJ.nlevels);
    status = 1;
    err = 0.f;
    ox = 0.f; oy = 0.f;
    for (int level = nl - 1; level >= 0; level--)
        lk_level(PI.lv[level], PJ.lv[level], win, level, nl - 1, max_count, eps2, px, py, ox, oy, status, err, ldsI, ldsD, lane, n_iter, n_setup, want_err);
}

// grid = (max points, batch), block = one wavefront
__global__ __launch_bounds__(64) void k_lk(const void* job_tab, size_t tab_stride)
{
    const LKJob& job = *reinterpret_cast<const LKJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blockIdx.y * tab_stride);
    const int n = LK_N_OF(job);
    if ((int)blockIdx.x >= n) return;
    const int pt = job.order ? ((gptr_i32)job.order)[blockIdx.x] : (int)blockIdx.x;  // launch slot -> point (LKJob::order)
    const int lane = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kmax = (job.win * job.win + 63) / 64;
    int* ldsD = reinterpret_cast<int*>(smem);
    short* ldsI = reinterpret_cast<short*>(smem + (size_t)kmax * 64 * 4);

    const float qx = ((gptr_f32)job.p_in)[2 * pt], qy = ((gptr_f32)job.p_in)[2 * pt + 1];
    const float px = __fsub_rn(__fmul_rn(qx, job.in_scale), job.in_off[0]);
    const float py = __fsub_rn(__fmul_rn(qy, job.in_scale), job.in_off[1]);

    float fx, fy, err;
    int st, n_iter = 0, n_setup = 0;
    lk_track(job.I, job.J, job.win, job.max_count, job.eps2, px, py, fx, fy, st, err, ldsI, ldsD, lane, n_iter, n_setup, job.err_out != nullptr);
    float fbe = 0.f;
    if (job.fbt >= 0.f) {
        float bx = 0.f, by = 0.f, e2;
        int st2 = 0;
        // forward status 0: `v = v & v2 & (fbe < fbt)` (KLT.py:50) is 0 whatever the backward pass finds, and the point returned is the forward one, so
        // the backward pass is skipped unless the caller asked for fbe itself (vh_pyr_lk); in the 4- / 8-tracks-per-wavefront kernels the dead tracks'
        // lanes sit out the pass (exec mask), and a wavefront whose tracks are all dead skips it
        if (st || job.fbe_out) lk_track(job.J, job.I, job.win, job.max_count, job.eps2, fx, fy, bx, by, st2, e2, ldsI, ldsD, lane, n_iter, n_setup, false);
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
            unsigned long long* st = job.stats + (size_t)(blockIdx.x & (VH_LK_STAT_SLOTS - 1)) * 16;  // (counters spread over VH_LK_STAT_SLOTS lines: see StreamWS::lk_stats)
            atomicAdd(&st[0], (unsigned long long)n_iter);
            atomicAdd(&st[1], (unsigned long long)n_setup);
        }
    }
}


// =================================================================================================================
// v2: strip kernel.  One lane handles a horizontal strip of 4 window samples: the 4x7 (set-up) / 2x5 (iteration)
// pixel block of a strip comes from 3 aligned dword loads per row + v_alignbyte, the bilinear / gradient dot products
// run on v_dot2_i32_i16 with packed int16 pairs, the template lives in LDS as lane-private packed int16 quads
// (conflict-free ds_read_b64), and the exact wave sums are DPP row reductions of 16-bit halves (order independent).
// Integer arithmetic is identical to the per-sample kernel above, so results are bit-identical.
// =================================================================================================================

__device__ __forceinline__ int dpp_row_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror
    return v;
}
// exact wave-wide sum of per-lane int32 partials (|v| < 2^31) as int64, uniform result
__device__ __forceinline__ long long wave_sum_i32_wide(int v)
{
    int lo = dpp_row_sum(v & 0xffff), hi = dpp_row_sum(v >> 16);
    // rows 0..3 -> lane 63: row_bcast:15 adds the previous row's total to rows 1 and 3, row_bcast:31 the total of rows 0+1 to rows 2 and 3
    // (two DPP adds and one v_readlane per half instead of four v_readlane and three scalar adds)
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false);
    const int lo_t = __builtin_amdgcn_readlane(lo, 63), hi_t = __builtin_amdgcn_readlane(hi, 63);
    return (long long)hi_t * 65536ll + (long long)lo_t;
}

// Two (or three) such sums at once, transposing while folding: v_permlane32_swap puts the upper half-wave of one vector beside the lower half-wave
// of another, so ONE add folds two vectors 64 -> 32 lanes; v_permlane16_swap does the same for 16-lane rows.  The four 16-bit halves of two values
// end up in the four rows of one register: 3 swaps + 3 adds + 4 DPP adds + 4 readlanes instead of 24 DPP adds + 6 readlanes (gfx950 only).
__device__ __forceinline__ int fold32_pair(int a, int b)  // lanes 0..31: a[l] + a[l+32], lanes 32..63: b[l-32] + b[l]
{
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    return (int)(r[0] + r[1]);
}
__device__ __forceinline__ int fold16_pair(int a, int b)  // rows 0/2: a[row] + a[row+1], rows 1/3: b[row-1] + b[row]
{
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    return (int)(r[0] + r[1]);
}
__device__ __forceinline__ void wave_sum2_i32_wide(int a, int b, long long& sa, long long& sb)
{
    const int x = fold32_pair(a & 0xffff, a >> 16), y = fold32_pair(b & 0xffff, b >> 16);
    const int z = dpp_row_sum(fold16_pair(x, y));  // rows: a.lo, b.lo, a.hi, b.hi
    sa = (long long)__builtin_amdgcn_readlane(z, 32) * 65536ll + (long long)__builtin_amdgcn_readlane(z, 0);
    sb = (long long)__builtin_amdgcn_readlane(z, 48) * 65536ll + (long long)__builtin_amdgcn_readlane(z, 16);
}
__device__ __forceinline__ long long wave_sum1_i32_wide_swap(int a)
{
    const int x = fold32_pair(a & 0xffff, a >> 16);
    const int z = dpp_row_sum(fold16_pair(x, x));  // rows 0,1: a.lo, rows 2,3: a.hi
    return (long long)__builtin_amdgcn_readlane(z, 32) * 65536ll + (long long)__builtin_amdgcn_readlane(z, 0);
}

// wave-wide sum when 16 lanes of partials provably fit int32 (one 4-sample strip per lane: 16 * 4 * 8160 * 4080 < 2^31):
// one DPP chain per value instead of two
__device__ __forceinline__ long long wave_sum_i32_rows(int v)
{
    const int r = dpp_row_sum(v);
    return (long long)__builtin_amdgcn_readlane(r, 0) + (long long)__builtin_amdgcn_readlane(r, 16) +
           (long long)__builtin_amdgcn_readlane(r, 32) + (long long)__builtin_amdgcn_readlane(r, 48);
}


// NR rows x 8 bytes starting at pixel (gx, gy): lo = bytes 0..3, hi = bytes 4..7 of every row (interior fast path)
typedef const unsigned __attribute__((address_space(1)))* gptr_u32;  // global (not flat) loads

template <int NR>
__device__ __forceinline__ void load_rows_fast(const ImgDesc& im, int gx, int gy, unsigned* lo, unsigned* hi)
{
    const uint8_t* base = im.p + (ptrdiff_t)gy * im.stride + gx;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(base + (ptrdiff_t)r * im.stride);
        const unsigned sh = (unsigned)(a & 3);
        gptr_u32 ap = (gptr_u32)(a - sh);
        const unsigned d0 = ap[0], d1 = ap[1], d2 = ap[2];
        lo[r] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        hi[r] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    }
}
// the same block through REFLECT_101 byte loads (windows that touch the border)
template <int NR, int NC>
__device__ __forceinline__ void load_rows_slow(const ImgDesc& im, int gx, int gy, unsigned* lo, unsigned* hi)
{
#pragma unroll
    for (int r = 0; r < NR; r++) {
        unsigned l = 0, h = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const unsigned v = (unsigned)pix_r(im, gx + c, gy + r);
            if (c < 4) l |= v << (8 * c);
            else h |= v << (8 * (c - 4));
        }
        lo[r] = l; hi[r] = h;
    }
}
__device__ __forceinline__ int byte_of(unsigned lo, unsigned hi, int c) { return c < 4 ? (int)((lo >> (8 * c)) & 0xff) : (int)((hi >> (8 * (c - 4))) & 0xff); }

struct StripWeights {
    unsigned wt, wb;  // packed (w00,w01), (w10,w11)
    int w00, w01, w10, w11;
};
__device__ __forceinline__ StripWeights strip_weights(const Win& w)
{
    StripWeights s;
    s.w00 = w.w00; s.w01 = w.w01; s.w10 = w.w10; s.w11 = w.w11;
    s.wt = pack16(w.w00, w.w01); s.wb = pack16(w.w10, w.w11);
    return s;
}

// packed horizontal neighbours (b_c, b_{c+1}), c = 0..3, of one row (lo = bytes 0..3, hi = byte 4..)
__device__ __forceinline__ void strip_row_pairs(unsigned lo, unsigned hi, unsigned* t)
{
    t[0] = __builtin_amdgcn_perm(0u, lo, 0x0c010c00u);
    t[1] = __builtin_amdgcn_perm(0u, lo, 0x0c020c01u);
    t[2] = __builtin_amdgcn_perm(0u, lo, 0x0c030c02u);
    t[3] = __builtin_amdgcn_perm(hi, lo, 0x0c040c03u);
}
// bilinear x32 samples of the 4 strip positions from the packed pairs of the top and bottom row
__device__ __forceinline__ void strip_bilinear_pairs(const unsigned* t, const unsigned* b, const StripWeights& w, unsigned& p01, unsigned& p23)
{
    // descale(acc, 9) = (acc + 256) >> 9 = upper half of (acc + 256) << 7  (acc < 2^22): one v_lshl_add per sample, one v_perm per pair
    int v[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
        v[c] = (dot2(b[c], w.wb, dot2_first(t[c], w.wt)) << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
    p01 = pack_hi16(v[0], v[1]);
    p23 = pack_hi16(v[2], v[3]);
}
// bilinear x32 samples of the 4 strip positions from 2 rows (lo,hi): returns packed pairs (v0,v1), (v2,v3)
__device__ __forceinline__ void strip_bilinear(const unsigned* lo, const unsigned* hi, const StripWeights& w, unsigned& p01, unsigned& p23)
{
    unsigned t[4], b[4];
    strip_row_pairs(lo[0], hi[0], t);
    strip_row_pairs(lo[1], hi[1], b);
    strip_bilinear_pairs(t, b, w, p01, p23);
}


// set-up of one strip from its 4x7 pixel block: template samples (x32), Scharr gradients, structure-tensor partials
// Interior windows: bilinear interpolation and the Scharr operator are both integer-linear, so
//   sum_k w_k * Scharr(P)[pos + k]  ==  Scharr(V)[pos]   with   V[pos] = sum_k w_k * P[pos + k]   (no rounding before the descale)
// V is needed on a 3 x 6 block for the 4 samples of a strip (18 x 2 v_dot2 on packed byte pairs); the template sample itself
// is descale(V, 9).  Exactly the same integers as interpolating the gradient image, at ~55 % of the instructions.
// the six packed byte pairs (c, c+1), c = 0..5, of one 8-byte patch row
__device__ __forceinline__ void setup_row_pairs(unsigned lo, unsigned hi, unsigned* pr)
{
    pr[0] = __builtin_amdgcn_perm(0u, lo, 0x0c010c00u);
    pr[1] = __builtin_amdgcn_perm(0u, lo, 0x0c020c01u);
    pr[2] = __builtin_amdgcn_perm(0u, lo, 0x0c030c02u);
    pr[3] = __builtin_amdgcn_perm(hi, lo, 0x0c040c03u);
    pr[4] = __builtin_amdgcn_perm(0u, hi, 0x0c010c00u);
    pr[5] = __builtin_amdgcn_perm(0u, hi, 0x0c020c01u);
}
// one row of V from the pairs of two consecutive patch rows
__device__ __forceinline__ void setup_v_row(const unsigned* top, const unsigned* bot, unsigned wt, unsigned wb, int* V)
{
#pragma unroll
    for (int c = 0; c < 6; c++) V[c] = dot2(bot[c], wb, dot2_first(top[c], wt));
}
// template samples, gradients and structure-tensor partials of one strip from its three V rows
__device__ __forceinline__ void setup_from_v(const int* V0, const int* V1, const int* V2, int cnt, uint2* tI, uint2* tX, uint2* tY, int slot,
                                             int& a11, int& a12, int& a22)
{
    // gradients x4 with the rounding folded in (column c of S carries 2^14 c, so S[c+2] - S[c] brings 4 * 2^13), descale + int16 packing by
    // v_perm of the upper halves, zeroing of the samples beyond the window edge by the selector (see setup_from_hg)
    int S4[6], D[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        S4[c] = mad24_v<40>(V1[c], mad24_s<12>(V0[c] + V2[c], c << W_BITS));  // every V fits 23 bits
        D[c] = V2[c] - V0[c];
    }
    int iv[4], ix[4], iy[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        iv[c] = (V1[c + 1] << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
        ix[c] = S4[c + 2] - S4[c];
        iy[c] = mad24_v<40>(D[c + 1], mad24_s<12>(D[c] + D[c + 2], 4 << (W_BITS - 1)));
    }
    const unsigned sel01 = cnt >= 2 ? 0x07060302u : 0x0c0c0302u, sel23 = cnt >= 4 ? 0x07060302u : (cnt == 3 ? 0x0c0c0302u : 0x0c0c0c0cu);
    const uint2 vI = make_uint2(__builtin_amdgcn_perm((unsigned)iv[1], (unsigned)iv[0], sel01), __builtin_amdgcn_perm((unsigned)iv[3], (unsigned)iv[2], sel23));
    const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ix[1], (unsigned)ix[0], sel01), __builtin_amdgcn_perm((unsigned)ix[3], (unsigned)ix[2], sel23));
    const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iy[1], (unsigned)iy[0], sel01), __builtin_amdgcn_perm((unsigned)iy[3], (unsigned)iy[2], sel23));
    tI[slot] = vI; tX[slot] = vX; tY[slot] = vY;
    a11 = dot2(vX.y, vX.y, dot2(vX.x, vX.x, a11));
    a12 = dot2(vX.y, vY.y, dot2(vX.x, vY.x, a12));
    a22 = dot2(vY.y, vY.y, dot2(vY.x, vY.x, a22));
}

// Rolling form for a vertical run of strips: per V row keep its horizontal differences H[c] = V[c+2] - V[c] and its
// horizontally smoothed values G[c] = 3 (V[c] + V[c+2]) + 10 V[c+1], c = 0..3.  Then (exact integer identities)
//   Scharr_x = 3 (H0 + H2) + 10 H1,   Scharr_y = G2 - G0,   so a new strip costs one H row and one G row, not a 3 x 6 block of S / D
// Both gradients are kept x4 with the descale rounding folded in, so that descale(., 14) + int16 packing is the upper half of the word
// (pack_hi16): G4 row g carries the addend 2^14 * g, hence G4[g+2] - G4[g] = 4 (G2 - G0) + 4 * 2^13; |4 * Scharr| < 2^29.
__device__ __forceinline__ void setup_hg_row(const int* V, int* H, int* G4, int g)
{
#pragma unroll
    for (int c = 0; c < 4; c++) {
        H[c] = V[c + 2] - V[c];
        G4[c] = mad24_v<40>(V[c + 1], mad24_s<12>(V[c] + V[c + 2], g << W_BITS));  // every V fits 22 bits
    }
}
// vmask: 0xffffffff for a strip inside the window, 0 for a strip of the last run that hangs below it (stored as zeros)
// sel01 / sel23: the v_perm selectors that pack the two sample pairs -- 0x07060302 keeps both upper halves, 0x0c0c0302 zeroes the second sample,
// 0x0c0c0c0c both: samples beyond the window edge and strips below the window cost no masking instruction
__device__ __forceinline__ void setup_from_hg(const int* H0, const int* H1, const int* H2, const int* G0, const int* G2, const int* Vmid, unsigned sel01,
                                              unsigned sel23, uint2* tI, uint2* tX, uint2* tY, int slot, int& a11, int& a12, int& a22)
{
    int iv[4], ix[4], iy[4];  // the wanted int16 in the upper half of each
#pragma unroll
    for (int c = 0; c < 4; c++) {
        iv[c] = (Vmid[c] << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
        ix[c] = mad24_v<40>(H1[c], mad24_s<12>(H0[c] + H2[c], 4 << (W_BITS - 1)));
        iy[c] = G2[c] - G0[c];
    }
    const uint2 vI = make_uint2(pack_hi16(iv[0], iv[1]), pack_hi16(iv[2], iv[3]));  // enters only through products with the (masked) gradients
    const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ix[1], (unsigned)ix[0], sel01), __builtin_amdgcn_perm((unsigned)ix[3], (unsigned)ix[2], sel23));
    const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iy[1], (unsigned)iy[0], sel01), __builtin_amdgcn_perm((unsigned)iy[3], (unsigned)iy[2], sel23));
    tI[slot] = vI; tX[slot] = vX; tY[slot] = vY;
    a11 = dot2(vX.y, vX.y, dot2(vX.x, vX.x, a11));
    a12 = dot2(vX.y, vY.y, dot2(vX.x, vY.x, a12));
    a22 = dot2(vY.y, vY.y, dot2(vY.x, vY.x, a22));
}

__device__ __forceinline__ void strip_setup_linear(const unsigned* lo, const unsigned* hi, const Win& w0, int cnt, uint2* tI, uint2* tX,
                                                   uint2* tY, int slot, int& a11, int& a12, int& a22)
{
    const unsigned wt = pack16(w0.w00, w0.w01), wb = pack16(w0.w10, w0.w11);
    unsigned pr[4][6];  // pr[r][c] = (byte c, byte c+1) of row r as a packed int16 pair
#pragma unroll
    for (int r = 0; r < 4; r++) setup_row_pairs(lo[r], hi[r], pr[r]);
    int V[3][6];
#pragma unroll
    for (int r = 0; r < 3; r++) setup_v_row(pr[r], pr[r + 1], wt, wb, V[r]);
    setup_from_v(V[0], V[1], V[2], cnt, tI, tX, tY, slot, a11, a12, a22);
}

template <bool FAST>
__device__ __forceinline__ void strip_setup(const unsigned* lo, const unsigned* hi, const Win& w0, const ImgDesc& I, int ipx, int ipy, int x,
                                            int y, int cnt, uint2* tI, uint2* tX, uint2* tY, int slot, int& a11, int& a12, int& a22)
{
    if constexpr (FAST) {
        strip_setup_linear(lo, hi, w0, cnt, tI, tX, tY, slot, a11, a12, a22);
        return;
    }
    constexpr bool fast = FAST;
            // vertical [3 10 3] smooth / [-1 0 1] difference for the two derivative rows, 7 columns
            int sm[2][7], df[2][7];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < 7; c++) {
                    const int t = byte_of(lo[r], hi[r], c), m = byte_of(lo[r + 1], hi[r + 1], c), b = byte_of(lo[r + 2], hi[r + 2], c);
                    sm[r][c] = (t + b) * 3 + m * 10;
                    df[r][c] = b - t;
                }
            int gx[2][5], gy[2][5];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    gx[r][c] = sm[r][c + 2] - sm[r][c];
                    gy[r][c] = (df[r][c] + df[r][c + 2]) * 3 + df[r][c + 1] * 10;
                    if (!fast) {  // the derivative image is constant 0 outside the level
                        const int ax = ipx + x + c, ay = ipy + y + r;
                        if (ax < 0 || ax >= I.w || ay < 0 || ay >= I.h) { gx[r][c] = 0; gy[r][c] = 0; }
                    }
                }
            int iv[4], ix[4], iy[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int q00 = byte_of(lo[1], hi[1], c + 1), q01 = byte_of(lo[1], hi[1], c + 2);
                const int q10 = byte_of(lo[2], hi[2], c + 1), q11 = byte_of(lo[2], hi[2], c + 2);
                // every factor fits 24 signed bits (|pixel| <= 255, |gradient| <= 4080, weights <= 2^14): full-rate v_mad_i32_i24
                iv[c] = vh_descale(__mul24(q00, w0.w00) + __mul24(q01, w0.w01) + __mul24(q10, w0.w10) + __mul24(q11, w0.w11), W_BITS - 5);
                ix[c] = vh_descale(__mul24(gx[0][c], w0.w00) + __mul24(gx[0][c + 1], w0.w01) + __mul24(gx[1][c], w0.w10) + __mul24(gx[1][c + 1], w0.w11), W_BITS);
                iy[c] = vh_descale(__mul24(gy[0][c], w0.w00) + __mul24(gy[0][c + 1], w0.w01) + __mul24(gy[1][c], w0.w10) + __mul24(gy[1][c + 1], w0.w11), W_BITS);
                if (c >= cnt) { iv[c] = 0; ix[c] = 0; iy[c] = 0; }
            }
            const uint2 vI = make_uint2(pack16(iv[0], iv[1]), pack16(iv[2], iv[3]));
            const uint2 vX = make_uint2(pack16(ix[0], ix[1]), pack16(ix[2], ix[3]));
            const uint2 vY = make_uint2(pack16(iy[0], iy[1]), pack16(iy[2], iy[3]));
            tI[slot] = vI; tX[slot] = vX; tY[slot] = vY;
            a11 = dot2(vX.y, vX.y, dot2(vX.x, vX.x, a11));
            a12 = dot2(vX.y, vY.y, dot2(vX.x, vY.x, a12));
            a22 = dot2(vY.y, vY.y, dot2(vY.x, vY.x, a22));
}

template <int WIN_T>
__device__ void lk_level_strip(const ImgDesc I, const ImgDesc J, int win_rt, int level, int top_level, int max_count, double eps2,
                               float p0x, float p0y, float& nxo, float& nyo, int& status, float& err, uint2* tI, uint2* tX, uint2* tY,
                               int lane, int& n_iter, int& n_setup, bool want_err)
{
    const int win = WIN_T ? WIN_T : win_rt;
    const int spr = (win + 3) >> 2;   // strips per window row
    const int nstrips = spr * win;
    const float half = (float)(win - 1) * 0.5f;
    const float lscale = __uint_as_float((unsigned)(127 - level) << 23);  // 2^-level exactly = (float)(1. / (1 << level)), without the f64 division
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
    const Win w0 = bilinear_weights(__fsub_rn(px, (float)ipx), __fsub_rn(py, (float)ipy));
    n_setup++;
    // strip coordinates of this lane: s = lane + 64 k -> (row, 4*col); advanced incrementally
    const int j_first = lane % spr, y_first = lane / spr;
    const int jinc = 64 % spr, yinc = 64 / spr;

    int a11 = 0, a12 = 0, a22 = 0;
    {
        // interior window: the patch [ipx-1, ipx+win+1] x [ipy-1, ipy+win+1] lies inside the level (V identity).  The aligned 12-byte row reads of a
        // strip (and the padding samples of the last strip) may run past either end of a row: harmless inside the level, excluded where they would
        // leave it (first row to the left, last row to the right)
        const bool fast = ipx >= 1 && ipy >= 1 && ipx + win + 3 <= I.w && ipy + win + 2 <= I.h && !(ipy == 1 && ipx < 4) &&
                          !(ipy + win + 2 == I.h && ipx + 4 * spr + 7 > I.w);
        if constexpr (WIN_T != 0 && ((((WIN_T + 3) >> 2) * WIN_T + 63) / 64) <= 2) {
            constexpr int SPR = (WIN_T + 3) >> 2, NS = SPR * WIN_T, KMAX = (NS + 63) / 64, G = KMAX < 3 ? KMAX : 3;
#pragma unroll
            for (int kb = 0; kb < KMAX; kb += G) {
                unsigned lo[G][4], hi[G][4];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int k = kb + g, s = lane + 64 * k;
                    if (k < KMAX && s < NS) {
                        const int y = s / SPR, x = 4 * (s - y * SPR);
                        if (fast) load_rows_fast<4>(I, ipx + x - 1, ipy + y - 1, lo[g], hi[g]);
                        else load_rows_slow<4, 7>(I, ipx + x - 1, ipy + y - 1, lo[g], hi[g]);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int k = kb + g, s = lane + 64 * k;
                    if (k < KMAX && s < NS) {
                        const int y = s / SPR, x = 4 * (s - y * SPR), cnt = min(4, WIN_T - x);
                        if (fast) strip_setup<true>(lo[g], hi[g], w0, I, ipx, ipy, x, y, cnt, tI, tX, tY, k * 64 + lane, a11, a12, a22);
                        else strip_setup<false>(lo[g], hi[g], w0, I, ipx, ipy, x, y, cnt, tI, tX, tY, k * 64 + lane, a11, a12, a22);
                    }
                }
            }
        } else {
            int j = j_first, y = y_first, k = 0;
            for (int s = lane; s < nstrips; s += 64, k++) {
                const int x = 4 * j, cnt = min(4, win - x);
                unsigned lo[4], hi[4];
                if (fast) {
                    load_rows_fast<4>(I, ipx + x - 1, ipy + y - 1, lo, hi);
                    strip_setup<true>(lo, hi, w0, I, ipx, ipy, x, y, cnt, tI, tX, tY, k * 64 + lane, a11, a12, a22);
                } else {
                    load_rows_slow<4, 7>(I, ipx + x - 1, ipy + y - 1, lo, hi);
                    strip_setup<false>(lo, hi, w0, I, ipx, ipy, x, y, cnt, tI, tX, tY, k * 64 + lane, a11, a12, a22);
                }
                j += jinc; y += yinc;
                if (j >= spr) { j -= spr; y++; }
            }
        }
    }
    constexpr bool ROWSAFE = WIN_T != 0 && ((WIN_T + 3) >> 2) * WIN_T <= 64;  // one strip per lane
    const long long sA11 = ROWSAFE ? wave_sum_i32_rows(a11) : wave_sum_i32_wide(a11);
    const long long sA12 = ROWSAFE ? wave_sum_i32_rows(a12) : wave_sum_i32_wide(a12);
    const long long sA22 = ROWSAFE ? wave_sum_i32_rows(a22) : wave_sum_i32_wide(a22);
    const float A11 = __fmul_rn(i64_to_f32(sA11), LK_FLT_SCALE), A12 = __fmul_rn(i64_to_f32(sA12), LK_FLT_SCALE), A22 = __fmul_rn(i64_to_f32(sA22), LK_FLT_SCALE);
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
    for (int it = 0; it < max_count; it++) {
        const int inx = vh_floor(nx), iny = vh_floor(ny);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(nx, (float)inx), __fsub_rn(ny, (float)iny)));
        const bool fast = inx >= 0 && iny >= 0 && inx + win + 2 <= J.w && iny + win + 1 <= J.h && !(iny == 0 && inx < 3) &&
                          !(iny + win + 1 == J.h && inx + 4 * spr + 8 > J.w);  // see the set-up
        n_iter++;
        int b1 = 0, b2 = 0;
        if constexpr (WIN_T != 0 && ((((WIN_T + 3) >> 2) * WIN_T + 63) / 64) <= 2) {
            // small compile-time window: fully unrolled, all loads of a group of strips issued before the first use so a
            // Newton iteration pays the memory latency once per group instead of once per strip
            constexpr int SPR = (WIN_T + 3) >> 2, NS = SPR * WIN_T, KMAX = (NS + 63) / 64, G = KMAX < 6 ? KMAX : 6;
#pragma unroll
            for (int kb = 0; kb < KMAX; kb += G) {
                unsigned lo[G][2], hi[G][2];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int k = kb + g, s = lane + 64 * k;
                    if (k < KMAX && s < NS) {
                        const int y = s / SPR, j = s - y * SPR;
                        if (fast) load_rows_fast<2>(J, inx + 4 * j, iny + y, lo[g], hi[g]);
                        else load_rows_slow<2, 5>(J, inx + 4 * j, iny + y, lo[g], hi[g]);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int k = kb + g, s = lane + 64 * k;
                    if (k < KMAX && s < NS) {
                        unsigned p01, p23;
                        strip_bilinear(lo[g], hi[g], w, p01, p23);
                        const uint2 vI = tI[k * 64 + lane], vX = tX[k * 64 + lane], vY = tY[k * 64 + lane];
                        const unsigned d01 = __builtin_bit_cast(unsigned, as_s2(p01) - as_s2(vI.x));
                        const unsigned d23 = __builtin_bit_cast(unsigned, as_s2(p23) - as_s2(vI.y));
                        b1 = dot2(d23, vX.y, dot2(d01, vX.x, b1));
                        b2 = dot2(d23, vY.y, dot2(d01, vY.x, b2));
                    }
                }
            }
        } else {
            int j = j_first, y = y_first, k = 0;
            for (int s = lane; s < nstrips; s += 64, k++) {
                unsigned lo[2], hi[2];
                if (fast) load_rows_fast<2>(J, inx + 4 * j, iny + y, lo, hi);
                else load_rows_slow<2, 5>(J, inx + 4 * j, iny + y, lo, hi);
                unsigned p01, p23;
                strip_bilinear(lo, hi, w, p01, p23);
                const uint2 vI = tI[k * 64 + lane], vX = tX[k * 64 + lane], vY = tY[k * 64 + lane];
                const unsigned d01 = __builtin_bit_cast(unsigned, as_s2(p01) - as_s2(vI.x));
                const unsigned d23 = __builtin_bit_cast(unsigned, as_s2(p23) - as_s2(vI.y));
                // unused samples of the last strip of a row carry Ix = Iy = 0, so they add nothing
                b1 = dot2(d23, vX.y, dot2(d01, vX.x, b1));
                b2 = dot2(d23, vY.y, dot2(d01, vY.x, b2));
                j += jinc; y += yinc;
                if (j >= spr) { j -= spr; y++; }
            }
        }
        const long long sb1 = ROWSAFE ? wave_sum_i32_rows(b1) : wave_sum_i32_wide(b1);
        const long long sb2 = ROWSAFE ? wave_sum_i32_rows(b2) : wave_sum_i32_wide(b2);
        const float fb1 = __fmul_rn(i64_to_f32(sb1), LK_FLT_SCALE), fb2 = __fmul_rn(i64_to_f32(sb2), LK_FLT_SCALE);
        const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
        const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
        nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
        nxo = __fadd_rn(nx, half); nyo = __fadd_rn(ny, half);
        if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= eps2) break;
        if (it > 0 && fabsf(__fadd_rn(dx, pdx)) < 0.01f && fabsf(__fadd_rn(dy, pdy)) < 0.01f) {
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
        if (!want_err) return;  // err is discarded by the caller (KLT.py:83 `pa, v, _`): only the bounds rule matters
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(fx, (float)inx), __fsub_rn(fy, (float)iny)));
        const bool fast = inx >= 0 && iny >= 0 && inx + win + 2 <= J.w && iny + win + 1 <= J.h && !(iny == 0 && inx < 3) &&
                          !(iny + win + 1 == J.h && inx + 4 * spr + 8 > J.w);  // see the set-up
        int se = 0;
        int j = j_first, y = y_first, k = 0;
        for (int s = lane; s < nstrips; s += 64, k++) {
            const int cnt = min(4, win - 4 * j);
            unsigned lo[2], hi[2];
            if (fast) load_rows_fast<2>(J, inx + 4 * j, iny + y, lo, hi);
            else load_rows_slow<2, 5>(J, inx + 4 * j, iny + y, lo, hi);
            unsigned p01, p23;
            strip_bilinear(lo, hi, w, p01, p23);
            const uint2 vI = tI[k * 64 + lane];
            const short2v d01 = as_s2(p01) - as_s2(vI.x), d23 = as_s2(p23) - as_s2(vI.y);
            const int d[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
            for (int c = 0; c < 4; c++) se += c < cnt ? (d[c] < 0 ? -d[c] : d[c]) : 0;
            j += jinc; y += yinc;
            if (j >= spr) { j -= spr; y++; }
        }
        const long long sse = ROWSAFE ? wave_sum_i32_rows(se) : wave_sum_i32_wide(se);
        err = __fmul_rn(i64_to_f32(sse), __fdiv_rn(1.f, (float)(32 * win * win)));
    }
}

template <int WIN_T>
__device__ void lk_track_strip(const PyrDesc& PI, const PyrDesc& PJ, int win, int max_count, double eps2, float px, float py, float& ox,
                               float& oy, int& status, float& err, uint2* tI, uint2* tX, uint2* tY, int lane, int& n_iter, int& n_setup, bool want_err)
{
    const int nl = min(PI.nlevels, PJ.nlevels);
    status = 1;
    err = 0.f;
    ox = 0.f; oy = 0.f;
    for (int level = nl - 1; level >= 0; level--)
        lk_level_strip<WIN_T>(PI.lv[level], PJ.lv[level], win, level, nl - 1, max_count, eps2, px, py, ox, oy, status, err, tI, tX, tY, lane,
                              n_iter, n_setup, want_err);
}


// XCD-aware, stream-interleaved block order of the batched LK launches (round 4).  The dispatcher deals consecutive workgroups (linear id
// L = y gridDim.x + x) round-robin to the 8 XCDs, in order.  With y = stream and x = track block, a stream's tracks are spread over all eight L2s, each L2
// sees the pyramids of every stream in flight, and the streams run one after the other.  Re-indexed: streams are taken in sets of G (G gridDim.x
// consecutive ids), id q of a set -> stream q mod G of the set, track block q / G.  With G a multiple of 8 every workgroup of stream y has L mod 8 == y mod 8:
// a stream lives in ONE L2 (its pyramid levels are read again and again by neighbouring tracks, both directions, every level and iteration), and G / 8
// streams share an XCD at any time, which evens out their run times for the in-order dispatcher.  (A last set of m < G streams is dealt q mod m: valid,
// XCD-pure only if m is a multiple of 8.)  Pure re-indexing: every (x, y) is produced exactly once, results are unchanged.
// Measured (tools/exp/lko_group_sweep*.sh, C2, one box): coarse k_lk_o at 256 streams 632 / 1076 us (natural order) -> 524 / 935 us for G = 64 .. 256
// (G = 31 / 63: 565 / 960; G = 8, ONE stream per XCD: 1169 / 1369 -- the XCDs drift apart and the in-order dispatcher waits for the slowest); fine k_lk3
// 3830 -> 3661 us at G = 8, 3675 at 16, 3725 at 32.  Small batches gain most: 8 streams 13.2 k -> 18.2 k frames/s, 32 streams 26.4 k -> 34.4 k.
// Most of it is the INTERLEAVING (resident workgroups come from many streams instead of one or two, all at the same pyramid level of the same images);
// the same order rotated off the XCDs (LK_GRP_ROTATE) keeps 95-98 %: 8 streams 17.9 k against 18.3 k pinned, 256 streams 40.9 k against 41.6 k.
#ifndef LK_XCD_REMAP
#define LK_XCD_REMAP 1
#endif
#define LK_GRP_ROTATE 0x10000u  // flag in the group argument: stream (q + q / m) mod m instead of q mod m -- interleaved, but NOT pinned to an XCD
template <bool REMAP>
__device__ __forceinline__ void lk_block_xy(unsigned& bx, unsigned& by, unsigned grp = 8u)
{
    bx = blockIdx.x; by = blockIdx.y;
    const unsigned G = grp & 0xffffu;
    if (REMAP && LK_XCD_REMAP && gridDim.y > 1 && G > 1u) {
        const unsigned nx = gridDim.x, L = blockIdx.y * nx + blockIdx.x;
        const unsigned set = L / (G * nx), q = L - set * G * nx;
        const unsigned m = min(G, gridDim.y - set * G);
        const unsigned xq = q / m;
        by = set * G + ((q - xq * m) + ((grp & LK_GRP_ROTATE) ? xq : 0u)) % m;
        bx = xq;
    }
}
// group argument of a launch: sets of `g` streams; pinned from 64 streams on (8+ streams share an XCD: their run times even out), rotated below (a stream's
// workgroups visit every XCD, so a stream that carries more tracks than the others cannot hold one XCD -- and with it the in-order dispatcher -- back)
static inline unsigned lk_group_arg(unsigned g, int batch) { return g | (batch < 64 ? LK_GRP_ROTATE : 0u); }

template <int WIN_T>
__global__ __launch_bounds__(64) void k_lk_strip(const void* job_tab, size_t tab_stride)
{
    unsigned blk_x, blk_y;
    lk_block_xy<false>(blk_x, blk_y);
    const LKJob& job = *reinterpret_cast<const LKJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blk_y * tab_stride);
    const int n = LK_N_OF(job);
    if ((int)blk_x >= n) return;
    const int pt = job.order ? ((gptr_i32)job.order)[blk_x] : (int)blk_x;  // launch slot -> point (LKJob::order)
    const int lane = threadIdx.x;
    const int win = WIN_T ? WIN_T : job.win;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kmax = (((win + 3) >> 2) * win + 63) / 64;
    uint2* tI = reinterpret_cast<uint2*>(smem);
    uint2* tX = tI + kmax * 64;
    uint2* tY = tX + kmax * 64;

    const float qx = ((gptr_f32)job.p_in)[2 * pt], qy = ((gptr_f32)job.p_in)[2 * pt + 1];
    const float px = __fsub_rn(__fmul_rn(qx, job.in_scale), job.in_off[0]);
    const float py = __fsub_rn(__fmul_rn(qy, job.in_scale), job.in_off[1]);

    float fx, fy, err;
    int st, n_iter = 0, n_setup = 0;
    lk_track_strip<WIN_T>(job.I, job.J, win, job.max_count, job.eps2, px, py, fx, fy, st, err, tI, tX, tY, lane, n_iter, n_setup, job.err_out != nullptr);
    float fbe = 0.f;
    if (job.fbt >= 0.f) {
        float bx = 0.f, by = 0.f, e2;
        int st2 = 0;
        if (st || job.fbe_out) lk_track_strip<WIN_T>(job.J, job.I, win, job.max_count, job.eps2, fx, fy, bx, by, st2, e2, tI, tX, tY, lane, n_iter, n_setup, false);
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
            unsigned long long* st = job.stats + (size_t)(blk_x & (VH_LK_STAT_SLOTS - 1)) * 16;  // (counters spread over VH_LK_STAT_SLOTS lines: see StreamWS::lk_stats)
            atomicAdd(&st[0], (unsigned long long)n_iter);
            atomicAdd(&st[1], (unsigned long long)n_setup);
        }
    }
}


// =================================================================================================================
// v3: LDS-staged kernel for the two windows the tracker uses (15x15 coarse, 51x51 fine).
// One workgroup of NW wavefronts per track.  Per level and direction the template patch ((WIN+3)^2 pixels of the
// previous image) and a search region ((WIN+1+2M)^2 pixels of the next image around the predicted position) are staged
// into LDS with ONE batch of global loads (REFLECT_101 applied while staging, so the compute loops have no border
// cases); set-up, every Newton iteration and the err pass then read LDS only.  The search region is re-staged if the
// window drifts more than M pixels.  Strip arithmetic, exact sums and control flow are those of the kernels above
// (bit-identical results); with NW > 1 the per-wave sums are combined through LDS.
// =================================================================================================================
template <int WIN, int NW, int M>
struct LK3 {
    static constexpr int T = 64 * NW;
    static constexpr int SPR = (WIN + 3) >> 2;
    static constexpr int NS = SPR * WIN;
    // lane -> (strip column j, run q): every lane owns K vertically consecutive strips of one column, so consecutive
    // strips share patch rows, byte pairs and V rows in registers (a strip's set-up needs 1 new V row instead of 3, a
    // Newton iteration 1 new search row instead of 2)
    // SPLIT (one wavefront, 51x51): 13 strip columns x 4 runs fill 52 lanes only, and 13 strips per lane cover 52 rows.  Instead every lane owns 11
    // strips: lanes 0..51 (column, run) as before with runs of 11 rows = rows 0..43; the 7 rows left (44..50) go to lanes 52..63 -- lane 52+c takes
    // column c < 12 in its first KA = 7 strips; the 7 strips of column 12 go ONE each to the last strip slot of lanes 52..58, whose slots 7..9 walk the
    // three rows above it so that the rolling state (two V rows in the set-up, one row of byte pairs in an iteration) is valid when slot 10 needs it.
    // Slots that hold no strip are kept as zeros by their selector (below).  663 strips in 64 x 11 slots instead of 52 x 13: -2 strips per lane.
    static constexpr bool SPLIT = (NW == 1 && WIN == 51);
    static constexpr int RUNS = SPLIT ? 4 : T / SPR;
    static constexpr int K = SPLIT ? 11 : (WIN + RUNS - 1) / RUNS;
    static constexpr int KA = SPLIT ? WIN - RUNS * K : K;  // strips of a lane's first segment
    static_assert(!SPLIT || (KA == 7 && SPR == 13 && RUNS * SPR + SPR - 1 <= T && KA <= T - RUNS * SPR && K - KA == 4), "split lane mapping");
    static constexpr int PI_ROWS = WIN + 3;                                   // template patch rows
    static constexpr int PI_PITCH = ((4 * (SPR - 1) + 8 + 3) / 4) * 4;        // bytes read per patch row, dword multiple
    static constexpr int RJ = WIN + 1 + 2 * M;                                // search region rows / cols
    static constexpr int PJ_WIDTH = ((2 * M + 4 * (SPR - 1)) >> 2) * 4 + 8;   // a row read takes two dwords
    static constexpr int PJ_PITCH = PJ_WIDTH + 4;                                // odd dword pitch (LDS banks)
    // The template of a lane's K strips (its Scharr gradients, 2 x 8 bytes per strip; the samples themselves enter only through the sums c, see
    // lk3_level) stays in REGISTERS for the whole level: LDS then
    // holds the two image regions only (~7.7 KB per workgroup for 51x51) and the VGPR file, not LDS, bounds the occupancy.  With the template in
    // LDS (24.5 KB per workgroup: 6 workgroups = 3 wavefronts per SIMD) the kernel lost 14 % per workgroup of occupancy taken away.
    // Every lane runs all K strips of its run with NO per-strip branch: strips below the window (the tail of the last run(s)) are computed from
    // whatever the padded buffers hold and kept as ZEROS, so they add nothing to any window sum in the set-up or in the Newton iterations.
    // (Branching on `y < WIN` per strip split the unrolled K loop into exec-masked blocks joined by register moves.)
    static constexpr int LANES = SPLIT ? T : RUNS * SPR;         // active lanes
    static constexpr int PAD_ROWS = SPLIT ? 0 : RUNS * K - WIN;  // window rows the last run(s) hang over the bottom edge
    static constexpr int OFF_PI = 0;
    static constexpr int OFF_PJ = OFF_PI + (PI_ROWS + PAD_ROWS) * PI_PITCH + 8;
    static constexpr int OFF_RED = ((OFF_PJ + (RJ + PAD_ROWS) * PJ_PITCH + 16 + 15) / 16) * 16;
    static constexpr int MAX_TPW = 8;                           // launch slots one workgroup may solve one after the other (k_lk3)
    static constexpr int OFF_RES = OFF_RED + 2 * NW * 4 * 8;   // their result records (8 dwords each), written out after the last one
    static constexpr int LDS_BYTES = OFF_RES + MAX_TPW * 32;
};

typedef const uint2 __attribute__((address_space(1)))* gptr_u32x2;
typedef int __attribute__((address_space(3)))* lds_i32;

// stage ROWS x PITCH bytes of image `im` starting at pixel (rx, ry) into LDS (region-aligned rows).  Compile-time
// extents: the loop is fully unrolled and every global load of the batch is issued before the first LDS store, so a
// staging costs one memory round trip.
template <int T, int ROWS, int WIDTH, int PITCH = WIDTH>  // WIDTH bytes of every row are staged, rows are PITCH bytes apart in LDS
__device__ __forceinline__ void stage_region(const ImgDesc& im, int rx, int ry, unsigned* dst, int tid)
{
    constexpr int DPR = WIDTH >> 2, LP = PITCH >> 2;
    // LPR lanes per region row, T / LPR rows per batch of loads: a lane keeps its dword column, so the address of batch `it` is the first one
    // plus the wave-uniform it * RPI * stride (no per-element division, one 32-bit add per load; hipcc's form of the flat index q / DPR cost
    // ~12 VALU instructions per element)
    constexpr int LPR = DPR <= 4 ? 4 : DPR <= 8 ? 8 : DPR <= 16 ? 16 : 32;
    static_assert(DPR <= 32 && T % LPR == 0, "region rows wider than 32 dwords are not staged by this layout");
    constexpr int RPI = T / LPR, NIT = (ROWS + RPI - 1) / RPI;
    const bool fast = rx >= 0 && ry >= 0 && rx + WIDTH + 4 <= im.w && ry + ROWS <= im.h;
    if (fast) {
        const int l = tid & (LPR - 1), r0 = tid / LPR;
        const int d = min(l, DPR - 1);
        const unsigned bsh = (unsigned)(reinterpret_cast<uintptr_t>(im.p) & 3);
        const uint8_t* bp = im.p - bsh;  // dword aligned, wave uniform
        const unsigned off0 = (unsigned)(__mul24(ry + r0, im.stride) + rx + 4 * d) + bsh;
        // A batch steps RPI rows = a multiple of 4 bytes whatever the stride is, so a lane keeps its byte phase and its dword-aligned offset over the
        // whole staging: one 32-bit add of a wave-uniform step per load pair (was: add + 2 and)
        static_assert(RPI % 4 == 0, "a batch of rows must keep the byte phase of a lane");
        const unsigned sh = off0 & 3u, aoff = off0 & ~3u;
        unsigned d0[NIT], d1[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            // unconditional loads (row clamped in the last batch): a branch around a load makes hipcc wait vmcnt(0) per element
            gptr_u32 ap;
            if ((it + 1) * RPI <= ROWS) ap = (gptr_u32)(bp + (aoff + (unsigned)(it * RPI) * (unsigned)im.stride));  // scalar base + 32-bit lane offset
            else ap = (gptr_u32)(bp + ((off0 + (unsigned)__mul24(min(it * RPI, ROWS - 1 - r0), im.stride)) & ~3u));  // clamped lanes are not stored
            d0[it] = ap[0]; d1[it] = ap[1];
        }
        if (l < DPR) {
#pragma unroll
            for (int it = 0; it < NIT; it++)
                if ((it + 1) * RPI <= ROWS || r0 + it * RPI < ROWS) dst[(r0 + it * RPI) * LP + l] = __builtin_amdgcn_alignbyte(d1[it], d0[it], sh);
        }
    } else {
        constexpr int TOTAL = ROWS * DPR;
        for (int q = tid; q < TOTAL; q += T) {
            const int r = q / DPR, d = q - r * DPR;
            unsigned v = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) v |= (unsigned)pix_r(im, rx + 4 * d + c, ry + r) << (8 * c);
            dst[r * LP + d] = v;
        }
    }
}

// block-wide exact sum of NV per-lane int32 partials -> int64 totals valid in every thread
template <int NW, int NV, bool ROWSAFE>
__device__ __forceinline__ void block_sum_wide(const int* part, long long* tot, long long* red, int& phase, int wave)
{
    long long w[NV];
    if constexpr (!ROWSAFE && NV >= 2) {
        wave_sum2_i32_wide(part[0], part[1], w[0], w[1]);
        if constexpr (NV == 3) w[2] = wave_sum1_i32_wide_swap(part[2]);
    } else {
#pragma unroll
        for (int k = 0; k < NV; k++) w[k] = ROWSAFE ? wave_sum_i32_rows(part[k]) : wave_sum_i32_wide(part[k]);
    }
    if (NW == 1) {
#pragma unroll
        for (int k = 0; k < NV; k++) tot[k] = w[k];
        return;
    }
    long long* buf = red + phase * NW * 4;  // double buffered: one barrier per reduction
    phase ^= 1;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; k++) buf[wave * 4 + k] = w[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        long long s = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) s += buf[q * 4 + k];
        tot[k] = s;
    }
}

template <int WIN, int NW, int M>
__device__ __forceinline__ void lk3_level(const ImgDesc I, const ImgDesc J, int level, int top_level, int max_count, double eps2, float p0x, float p0y,
                          float& nxo, float& nyo, int& status, float& err, char* smem, int tid, int& phase, int& n_iter, int& n_setup, bool want_err)
{
    using C = LK3<WIN, NW, M>;
    // this lane's template gradients (registers: every index below is a compile-time constant).  The template SAMPLES are not kept: with
    // c = sum I * (Ix, Iy) over the lane's strips, sum (J - I) * Ix = sum J * Ix - c  exactly, so the Newton accumulators start at -c
    // (same per-lane partial sums as subtracting per sample; 12 registers and a v_pk_sub per pair less)
    uint2 tI[C::K], tX[C::K], tY[C::K];
    int cI[2] = {0, 0};
    unsigned* pI = reinterpret_cast<unsigned*>(smem + C::OFF_PI);
    unsigned* pJ = reinterpret_cast<unsigned*>(smem + C::OFF_PJ);
    long long* red = reinterpret_cast<long long*>(smem + C::OFF_RED);
    const int wave = tid >> 6;

    const float half = (float)(WIN - 1) * 0.5f;
    const float lscale = __uint_as_float((unsigned)(127 - level) << 23);  // 2^-level exactly = (float)(1. / (1 << level)), without the f64 division
    float px = __fmul_rn(p0x, lscale), py = __fmul_rn(p0y, lscale);
    float nx, ny;
    if (level == top_level) { nx = px; ny = py; }
    else { nx = __fmul_rn(nxo, 2.f); ny = __fmul_rn(nyo, 2.f); }
    nxo = nx; nyo = ny;

    px = __fsub_rn(px, half); py = __fsub_rn(py, half);
    const int ipx = vh_floor(px), ipy = vh_floor(py);
    if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
        if (level == 0) { status = 0; err = 0.f; }
        return;
    }
    const Win w0 = bilinear_weights(__fsub_rn(px, (float)ipx), __fsub_rn(py, (float)ipy));
    n_setup++;
    nx = __fsub_rn(nx, half); ny = __fsub_rn(ny, half);

    // one batch of global loads: template patch + search region around the predicted position
    int rjx = vh_floor(nx) - M, rjy = vh_floor(ny) - M;
    __syncthreads();  // previous users of the patch buffers are done
    stage_region<C::T, C::PI_ROWS, C::PI_PITCH>(I, ipx - 1, ipy - 1, pI, tid);
    stage_region<C::T, C::RJ, C::PJ_WIDTH, C::PJ_PITCH>(J, rjx, rjy, pJ, tid);
    __syncthreads();

    const bool inside_I = ipx >= 1 && ipy >= 1 && ipx + WIN + 1 <= I.w - 1 && ipy + WIN + 1 <= I.h - 1;
    // this lane's strips: slots 0..KA-1 = column jA, window rows rA + k; slots KA..K-1 = column jB, rows rB + k (one segment unless C::SPLIT).
    // selA / selB / selC: the v_perm selectors that pack (and mask) the gradient pairs of slots [0, KA), [KA, K-1) and K-1
    int jA, rA, jB, rB;
    bool lane_on;
    unsigned selA01, selA23, selB01, selB23, selC01, selC23;
    const auto sel01_of = [](int cnt) { return cnt >= 2 ? 0x07060302u : (cnt == 1 ? 0x0c0c0302u : 0x0c0c0c0cu); };
    const auto sel23_of = [](int cnt) { return cnt >= 4 ? 0x07060302u : (cnt == 3 ? 0x0c0c0302u : 0x0c0c0c0cu); };
    if constexpr (C::SPLIT) {
        constexpr int MAIN = C::RUNS * C::SPR;  // 52
        const bool main_lane = tid < MAIN;
        const int run = tid / C::SPR;
        jA = main_lane ? tid - run * C::SPR : tid - MAIN;
        rA = main_lane ? run * C::K : C::RUNS * C::K;
        const bool tail = !main_lane && tid < MAIN + C::KA;  // carries one strip of the last column in its last slot
        jB = main_lane ? jA : C::SPR - 1;
        rB = main_lane ? rA : (tail ? C::RUNS * C::K + (tid - MAIN) - (C::K - 1) : 0);
        lane_on = true;
        const int cntA = min(4, WIN - 4 * jA), cntL = WIN - 4 * (C::SPR - 1);
        selA01 = sel01_of(cntA); selA23 = sel23_of(cntA);
        selB01 = main_lane ? selA01 : 0x0c0c0c0cu; selB23 = main_lane ? selA23 : 0x0c0c0c0cu;
        selC01 = main_lane ? selA01 : (tail ? sel01_of(cntL) : 0x0c0c0c0cu); selC23 = main_lane ? selA23 : (tail ? sel23_of(cntL) : 0x0c0c0c0cu);
    } else {
        const int run = tid / C::SPR;
        jA = jB = tid - run * C::SPR;
        rA = rB = run * C::K;
        lane_on = run < C::RUNS;
        const int cnt = min(4, WIN - 4 * jA);
        selA01 = selB01 = selC01 = sel01_of(cnt);
        selA23 = selB23 = selC23 = sel23_of(cnt);
    }
    // (k is a compile-time constant wherever these are used: the strip loops are fully unrolled)
    const auto slot_col = [&](int k) { return k < C::KA ? jA : jB; };
    const auto slot_row = [&](int k) { return (k < C::KA ? rA : rB) + k; };
    const auto slot_sel = [&](int k, unsigned& s01, unsigned& s23) {
        if (C::SPLIT) {
            s01 = k < C::KA ? selA01 : (k < C::K - 1 ? selB01 : selC01);
            s23 = k < C::KA ? selA23 : (k < C::K - 1 ? selB23 : selC23);
        } else {
            // only the last PAD_ROWS strips of a run can hang below the window: the selector is a per-level lane constant everywhere else
            const bool below = k >= C::K - C::PAD_ROWS && rA + k >= WIN;
            s01 = below ? 0x0c0c0c0cu : selA01;
            s23 = below ? 0x0c0c0c0cu : selA23;
        }
    };
    const auto slot_live = [&](int k) { unsigned a, b; slot_sel(k, a, b); return a != 0x0c0c0c0cu; };  // the slot holds a strip of the window
    constexpr int slot_base = 0, slot_stride = 1;
    constexpr int PIP = C::PI_PITCH >> 2;
    int part[3] = {0, 0, 0};
    if (lane_on && inside_I) {
        // interior: rolling V rows (see strip_setup_linear): patch rows y .. y+3 feed strip y; one new row per strip
        const unsigned wt = pack16(w0.w00, w0.w01), wb = pack16(w0.w10, w0.w11);
        const unsigned* colA = pI + jA + rA * PIP;
        const unsigned* colB = pI + jB + rB * PIP;
        const auto patch_row = [&](int k, int dr) { return (k < C::KA ? colA : colB) + (k + dr) * PIP; };  // patch row slot_row(k) + dr, this lane's 8 bytes
        unsigned prB[6];
        int H0[4], H1[4], G0[4], G1[4], Vm[4];  // of V rows y, y+1 (Vm = the middle row's sample columns, for the template value)
        {
            unsigned pr0[6], prA[6];
            int V0[6], V1[6];
            const unsigned* r0 = patch_row(0, 0);
            setup_row_pairs(r0[0], r0[1], pr0);
            setup_row_pairs(r0[PIP], r0[PIP + 1], prA);
            setup_row_pairs(r0[2 * PIP], r0[2 * PIP + 1], prB);
            setup_v_row(pr0, prA, wt, wb, V0);
            setup_v_row(prA, prB, wt, wb, V1);
            setup_hg_row(V0, H0, G0, 0);
            setup_hg_row(V1, H1, G1, 1);
#pragma unroll
            for (int c = 0; c < 4; c++) Vm[c] = V1[c + 1];
        }
        // One patch row is read one strip ahead of its use and a scheduling barrier closes every strip: left alone, hipcc hoists the LDS reads of all
        // K strips to the top of the unrolled loop and interleaves the strips (248 VGPRs for K = 13)
        unsigned nlo, nhi;
        {
            const unsigned* rn = patch_row(0, 3);
            nlo = rn[0]; nhi = rn[1];
        }
#pragma unroll
        for (int k = 0; k < C::K; k++) {
            {   // no branch per strip: see LK3 (rows past the patch read the padding / the next buffer: harmless garbage, masked by the selector)
                const unsigned clo = nlo, chi = nhi;
                if (k + 1 < C::K) {
                    const unsigned* rn = patch_row(k + 1, 3);
                    nlo = rn[0]; nhi = rn[1];
                }
                unsigned prN[6];
                int V2[6], H2[4], G2[4];
                setup_row_pairs(clo, chi, prN);
                setup_v_row(prB, prN, wt, wb, V2);
                setup_hg_row(V2, H2, G2, k + 2);
                unsigned s01, s23;
                slot_sel(k, s01, s23);
                setup_from_hg(H0, H1, H2, G0, G2, Vm, s01, s23, tI, tX, tY, slot_base + k * slot_stride, part[0], part[1], part[2]);
                cI[0] = dot2(tI[k].y, tX[k].y, dot2(tI[k].x, tX[k].x, cI[0]));
                cI[1] = dot2(tI[k].y, tY[k].y, dot2(tI[k].x, tY[k].x, cI[1]));
#pragma unroll
                for (int c = 0; c < 4; c++) { H0[c] = H1[c]; H1[c] = H2[c]; G0[c] = G1[c]; G1[c] = G2[c]; Vm[c] = V2[c + 1]; }
#pragma unroll
                for (int c = 0; c < 6; c++) prB[c] = prN[c];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (lane_on) {
#pragma unroll
        for (int k = 0; k < C::K; k++) {
            const int y = slot_row(k), j = slot_col(k);
            if (!slot_live(k)) {  // slots without a strip are zeros (the iterations read every slot)
                const uint2 z = make_uint2(0u, 0u);
                tI[slot_base + k * slot_stride] = z; tX[slot_base + k * slot_stride] = z; tY[slot_base + k * slot_stride] = z;
            } else {
                unsigned lo[4], hi[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned* row = pI + (y + r) * PIP + j;
                    lo[r] = row[0]; hi[r] = row[1];
                }
                strip_setup<false>(lo, hi, w0, I, ipx, ipy, 4 * j, y, min(4, WIN - 4 * j), tI, tX, tY, slot_base + k * slot_stride, part[0], part[1], part[2]);
                cI[0] = dot2(tI[k].y, tX[k].y, dot2(tI[k].x, tX[k].x, cI[0]));
                cI[1] = dot2(tI[k].y, tY[k].y, dot2(tI[k].x, tY[k].x, cI[1]));
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the K strips of this (rare) border path sequential: register pressure
        }
    }
    long long sA[3];
    block_sum_wide<NW, 3, (C::K == 1 && C::T == 64)>(part, sA, red, phase, wave);
    const float A11 = __fmul_rn(i64_to_f32(sA[0]), LK_FLT_SCALE), A12 = __fmul_rn(i64_to_f32(sA[1]), LK_FLT_SCALE), A22 = __fmul_rn(i64_to_f32(sA[2]), LK_FLT_SCALE);
    float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
    const float dA = __fsub_rn(A11, A22);
    const float disc = __fadd_rn(__fmul_rn(dA, dA), __fmul_rn(__fmul_rn(4.f, A12), A12));
    const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), vh_sqrtf(disc)), (float)(2 * WIN * WIN));
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
        if (level == 0) status = 0;
        return;
    }
    D = __fdiv_rn(1.f, D);
    const int ncI[2] = {-cI[0], -cI[1]};
    constexpr int PJP = C::PJ_PITCH >> 2;
    unsigned lcA = (unsigned)(C::OFF_PJ + 4 * (rA * PJP + jA)), lcB = (unsigned)(C::OFF_PJ + 4 * (rB * PJP + jB));  // (((inx - rjx) + 4 j) >> 2 = ((inx - rjx) >> 2) + j)
    // opaque to the compiler, once per level.  NOT `asm volatile`: a volatile asm counts as a possible store to any memory, and every load of the
    // pyramid descriptors behind it turns from a scalar load into a per-lane global load (round 5: +17 vector loads, -17 scalar loads and +48 VALU
    // instructions per track, +25 % wait cycles, +4-5 % kernel time -- what made every earlier form of this change slower than the code it shortened)
    asm("" : "+v"(lcA), "+v"(lcB));

    // packed byte pairs of window row y (strip column j) of the staged search region at window origin (inx, iny)
    // The strip starts at byte `off` of the staged row: its 5 bytes lie inside the two dwords at off >> 2, and the byte pair (c, c+1) is ONE
    // v_perm of those two dwords with the selector 0x0c000c00 + (c + sh) * 0x00010001 + 0x00010000, sh = off & 3 (wave uniform: no alignbyte)
    auto region_row_pairs = [&](int inx, int iny, int j, int y, unsigned* t) {
        const int off = (inx - rjx) + 4 * j;
        const unsigned sel0 = 0x0c010c00u + (unsigned)__builtin_amdgcn_readfirstlane((inx - rjx) & 3) * 0x00010001u;
        const unsigned* row = pJ + (iny - rjy + y) * (C::PJ_PITCH >> 2) + (off >> 2);
        const unsigned d0 = row[0], d1 = row[1];
#pragma unroll
        for (int c = 0; c < 4; c++) t[c] = __builtin_amdgcn_perm(d1, d0, sel0 + (unsigned)c * 0x00010001u);
    };
    auto region_holds = [&](int inx, int iny) { return inx >= rjx && iny >= rjy && inx + WIN + 1 <= rjx + C::RJ && iny + WIN + 1 <= rjy + C::RJ; };
    auto restage = [&](int inx, int iny) {
        rjx = inx - M; rjy = iny - M;
        __syncthreads();
        stage_region<C::T, C::RJ, C::PJ_WIDTH, C::PJ_PITCH>(J, rjx, rjy, pJ, tid);
        __syncthreads();
    };

    float pdx = 0.f, pdy = 0.f;
    for (int it = 0; it < max_count; it++) {
        const int inx = vh_floor(nx), iny = vh_floor(ny);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        if (!region_holds(inx, iny)) restage(inx, iny);
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(nx, (float)inx), __fsub_rn(ny, (float)iny)));
        n_iter++;
        int b[2] = {ncI[0], ncI[1]};  // (lanes without strips keep -c: the first strip of the others starts from it without a copy)
        if (lane_on) {
            // rows are read one strip ahead of their use, a scheduling barrier closes every strip (see the set-up loop)
            const unsigned sel0 = 0x0c010c00u + (unsigned)__builtin_amdgcn_readfirstlane((inx - rjx) & 3) * 0x00010001u;
            // address of the lane's first search row = (lane constant, fixed for the level) + (wave-uniform offset of this iteration's window origin); the
            // lane constants were made opaque at level entry, so the compiler keeps ONE base register per column and puts the row step
            // (k + dr) * PJ_PITCH <= 816 bytes into the offset fields of ds_read2_b32 instead of folding OFF_PJ into every access (one v_add per row)
            const int uo = __builtin_amdgcn_readfirstlane(4 * ((iny - rjy) * PJP + ((inx - rjx) >> 2)));
            const unsigned* rowA = reinterpret_cast<const unsigned*>(smem + (lcA + (unsigned)uo));
            const unsigned* rowB = reinterpret_cast<const unsigned*>(smem + (lcB + (unsigned)uo));
            const auto region_row = [&](int k, int dr) { return (k < C::KA ? rowA : rowB) + (k + dr) * PJP; };  // search row slot_row(k) + dr
            unsigned top[4];
            {
                const unsigned* row = region_row(0, 0);
                const unsigned d0 = row[0], d1 = row[1];
#pragma unroll
                for (int c = 0; c < 4; c++) top[c] = __builtin_amdgcn_perm(d1, d0, sel0 + (unsigned)c * 0x00010001u);
            }
            unsigned n0 = region_row(0, 1)[0], n1 = region_row(0, 1)[1];
#pragma unroll
            for (int k = 0; k < C::K; k++) {
                {   // every slot, no branch: the gradients of slots without a strip are zeros (set-up)
                    const unsigned c0 = n0, c1 = n1;
                    if (k + 1 < C::K) {
                        n0 = region_row(k + 1, 1)[0];
                        n1 = region_row(k + 1, 1)[1];
                    }
                    unsigned bot[4], p01, p23;  // the bottom row of strip y is the top row of strip y+1
#pragma unroll
                    for (int c = 0; c < 4; c++) bot[c] = __builtin_amdgcn_perm(c1, c0, sel0 + (unsigned)c * 0x00010001u);
                    strip_bilinear_pairs(top, bot, w, p01, p23);
                    const int slot = slot_base + k * slot_stride;
                    const uint2 vX = tX[slot], vY = tY[slot];
                    if (k == 0) {  // the chains start at -c, which the next iteration needs again: three-address first link, no copy
                        b[0] = dot2(p23, vX.y, dot2_keep(p01, vX.x, ncI[0]));
                        b[1] = dot2(p23, vY.y, dot2_keep(p01, vY.x, ncI[1]));
                    } else {
                        b[0] = dot2(p23, vX.y, dot2(p01, vX.x, b[0]));
                        b[1] = dot2(p23, vY.y, dot2(p01, vY.x, b[1]));
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) top[c] = bot[c];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        long long sb[2];
        block_sum_wide<NW, 2, (C::K == 1 && C::T == 64)>(b, sb, red, phase, wave);
        const float fb1 = __fmul_rn(i64_to_f32(sb[0]), LK_FLT_SCALE), fb2 = __fmul_rn(i64_to_f32(sb[1]), LK_FLT_SCALE);
        const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
        const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
        nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
        nxo = __fadd_rn(nx, half); nyo = __fadd_rn(ny, half);
        if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= eps2) break;
        if (it > 0 && fabsf(__fadd_rn(dx, pdx)) < 0.01f && fabsf(__fadd_rn(dy, pdy)) < 0.01f) {
            nxo = __fsub_rn(nxo, __fmul_rn(dx, 0.5f));
            nyo = __fsub_rn(nyo, __fmul_rn(dy, 0.5f));
            break;
        }
        pdx = dx; pdy = dy;
    }

    if (status && level == 0) {
        const float fx = __fsub_rn(nxo, half), fy = __fsub_rn(nyo, half);
        const int inx = vh_floor(fx), iny = vh_floor(fy);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) { status = 0; return; }
        if (!want_err) return;  // err is discarded by the caller (KLT.py:83 `pa, v, _`): only the bounds rule matters
        if (!region_holds(inx, iny)) restage(inx, iny);
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(fx, (float)inx), __fsub_rn(fy, (float)iny)));
        int se[1] = {0};
        if (lane_on) {
#pragma unroll
            for (int k = 0; k < C::K; k++) {
                if (slot_live(k)) {
                    const int y = slot_row(k), j = slot_col(k), cnt = min(4, WIN - 4 * j);
                    unsigned top[4], bot[4], p01, p23;
                    region_row_pairs(inx, iny, j, y, top);
                    region_row_pairs(inx, iny, j, y + 1, bot);
                    strip_bilinear_pairs(top, bot, w, p01, p23);
                    // template samples of strip y again from the staged patch (rows y+1, y+2; byte columns 1..5 of the lane's 8 bytes)
                    uint2 vI;
                    {
                        const unsigned* r1 = pI + (y + 1) * (C::PI_PITCH >> 2) + j;
                        const unsigned* r2 = r1 + (C::PI_PITCH >> 2);
                        unsigned tp[4], bt[4];
                        strip_row_pairs(__builtin_amdgcn_alignbyte(r1[1], r1[0], 1), r1[1] >> 8, tp);
                        strip_row_pairs(__builtin_amdgcn_alignbyte(r2[1], r2[0], 1), r2[1] >> 8, bt);
                        strip_bilinear_pairs(tp, bt, strip_weights(w0), vI.x, vI.y);
                    }
                    const short2v d01 = as_s2(p01) - as_s2(vI.x), d23 = as_s2(p23) - as_s2(vI.y);
                    const int d[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
                    for (int c = 0; c < 4; c++) se[0] += c < cnt ? (d[c] < 0 ? -d[c] : d[c]) : 0;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        long long sse[1];
        block_sum_wide<NW, 1, (C::K == 1 && C::T == 64)>(se, sse, red, phase, wave);
        err = __fmul_rn(i64_to_f32(sse[0]), __fdiv_rn(1.f, (float)(32 * WIN * WIN)));
    }
}

template <int WIN, int NW, int M>
__device__ __forceinline__ void lk3_track(const PyrDesc& PI, const PyrDesc& PJ, int max_count, double eps2, float px, float py, float& ox, float& oy,
                          int& status, float& err, char* smem, int tid, int& phase, int& n_iter, int& n_setup, bool want_err)
{
    const int nl = min(PI.nlevels, PJ.nlevels);
    status = 1;
    err = 0.f;
    ox = 0.f; oy = 0.f;
    for (int level = nl - 1; level >= 0; level--)
        lk3_level<WIN, NW, M>(PI.lv[level], PJ.lv[level], level, nl - 1, max_count, eps2, px, py, ox, oy, status, err, smem, tid, phase, n_iter,
                              n_setup, want_err);
}

// (forcing 5 or 6 workgroups per CU through the second launch bound spills and measured 3-8 % slower; a 128-VGPR cap
// + a 3-pixel search margin to fit 7 workgroups of the 2-wave variant per CU: 11 spilled registers, 5 % slower: not used)
template <int WIN, int NW, int M>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 1 ? 3 : 4))) void k_lk3(const void* job_tab, size_t tab_stride, unsigned grp, unsigned tpw)
{
    unsigned blk_x0, blk_y0;
    lk_block_xy<true>(blk_x0, blk_y0, grp);
    const unsigned blk_x = (unsigned)__builtin_amdgcn_readfirstlane((int)blk_x0), blk_y = (unsigned)__builtin_amdgcn_readfirstlane((int)blk_y0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int phase = 0, tot_iter = 0, tot_setup = 0;
    unsigned ndone = 0;
    // A workgroup solves `tpw` consecutive launch slots one after the other (tracks that are neighbours in the launch order): fewer, longer-lived
    // workgroups.  Everything but the slot counter is re-derived per track (scalar loads of the job descriptor): values kept live across the loop cost
    // scalar registers the track code spills
#pragma unroll 1
    for (unsigned ti = 0; ti < tpw; ti++) {
        const LKJob& job = *reinterpret_cast<const LKJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blk_y * tab_stride);
        const int n = LK_N_OF(job);
        const unsigned slot = blk_x * tpw + ti;
        if ((int)slot >= n) break;
        const int tid = threadIdx.x;
        const int max_count = job.max_count;
        const double eps2 = job.eps2;
        const float fbt = job.fbt;
        // the point index and the start position are workgroup uniform and live until the epilogue: as SCALARS (the loads below come back in vector registers,
        // and hipcc then parks them in scratch for the length of the kernel: 7 spilled registers, 28 bytes of scratch traffic per lane and track)
        const int pt = __builtin_amdgcn_readfirstlane(job.order ? ((gptr_i32)job.order)[slot] : (int)slot);  // launch slot -> point (LKJob::order)

        const float qx = ((gptr_f32)job.p_in)[2 * pt], qy = ((gptr_f32)job.p_in)[2 * pt + 1];
        const float px = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(__fsub_rn(__fmul_rn(qx, job.in_scale), job.in_off[0]))));
        const float py = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(__fsub_rn(__fmul_rn(qy, job.in_scale), job.in_off[1]))));

        // forward pass, then (fbt >= 0) the backward pass from its result: ONE copy of the track code in a loop over the direction -- two inlined copies
        // doubled the kernel and recomputed the per-lane mapping / masks of lk3_level in each
        float fx = 0.f, fy = 0.f, err = 0.f, bx = 0.f, by = 0.f;
        int st = 0, st2 = 0, n_iter = 0, n_setup = 0;
        const int ndir = fbt >= 0.f ? 2 : 1;
        const bool want_err = job.err_out != nullptr;
    #pragma unroll 1
        for (int dir = 0; dir < ndir; dir++) {
            if (dir == 1 && !st && !job.fbe_out) break;  // forward status 0 (block uniform): the backward pass cannot change v or p (see k_lk)
            const PyrDesc& PA = dir ? job.J : job.I;
            const PyrDesc& PB = dir ? job.I : job.J;
            float ox, oy, e;
            int s;
            lk3_track<WIN, NW, M>(PA, PB, max_count, eps2, dir ? fx : px, dir ? fy : py, ox, oy, s, e, smem, tid, phase, n_iter, n_setup, want_err && dir == 0);
            if (dir == 0) { fx = ox; fy = oy; st = s; err = e; }
            else { bx = ox; by = oy; st2 = s; }
        }
        float fbe = 0.f;
        if (fbt >= 0.f) {
            const float ddx = __fsub_rn(px, bx), ddy = __fsub_rn(py, by);
            fbe = vh_sqrtf(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)));
            st = st && st2 && (fbe < fbt);
        }
        // the track's results wait in LDS: a global store inside this loop would make every descriptor load of the NEXT track a possibly-clobbered load,
        // i.e. a vector load instead of a scalar one (the same effect as the volatile asm of DESIGN.md section 9)
        if (tid == 0) {
            // (an explicit LDS pointer: through a generic one hipcc counts these stores as possible writes to the job descriptors too)
            lds_i32 rec = (lds_i32)(smem + LK3<WIN, NW, M>::OFF_RES) + 8 * ti;
            rec[0] = pt; rec[1] = __float_as_int(fx); rec[2] = __float_as_int(fy); rec[3] = st != 0;
            rec[4] = __float_as_int(err); rec[5] = __float_as_int(fbe);
        }
        tot_iter += n_iter; tot_setup += n_setup;
        ndone = ti + 1;
        // The LAST memory operation of the loop body must not be a plain store: hipcc's scalar-load test (is a load clobbered anywhere in the function?) takes
        // whatever definition reaches the loop header over the back edge at face value when it is a store -- LDS or not -- and only looks through fences
        // and barriers with alias analysis.  With the record stores last, every descriptor load of the track code became a vector load.
        __syncthreads();
    }
    const LKJob& job = *reinterpret_cast<const LKJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blk_y * tab_stride);
    if (threadIdx.x < ndone) {
        const int* rec = reinterpret_cast<const int*>(smem + LK3<WIN, NW, M>::OFF_RES) + 8 * threadIdx.x;
        const int pt = rec[0];
        const float fx = __int_as_float(rec[1]), fy = __int_as_float(rec[2]), err = __int_as_float(rec[4]), fbe = __int_as_float(rec[5]);
        const int st = rec[3];
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
    }
    if (threadIdx.x == 0 && job.stats && ndone) {
        unsigned long long* st = job.stats + (size_t)(blk_x & (VH_LK_STAT_SLOTS - 1)) * 16;  // (counters spread over VH_LK_STAT_SLOTS lines: see StreamWS::lk_stats)
        atomicAdd(&st[0], (unsigned long long)tot_iter);
        atomicAdd(&st[1], (unsigned long long)tot_setup);
    }
}

// =================================================================================================================
// v4: quarter-wave kernel for small windows (WIN <= 16, the 15x15 coarse stages).
// One DPP row (16 lanes) per track, 4 tracks per wavefront: lane r owns window row r and its (WIN+3)/4 strips, the
// template quads stay in registers (no LDS at all), the exact window sums are 16-lane DPP reductions, and what used to be
// wave-uniform scalar work (weights, the 2x2 solve, the stop rules) is one vector instruction stream serving 4 tracks.
// A one-wave-per-track kernel spends about half of its instructions of a 15x15 Newton iteration on that uniform part
// and on the wave reduction; here it is shared 4 ways.  Rows of the search window are loaded once per lane and handed
// to the lane above through DPP (the bilinear cell needs rows r and r+1).  Same integers, same float sequence per
// track as the kernels above: bit-identical results.
// =================================================================================================================
__device__ __forceinline__ long long row16_sum_wide(int v)
{
    const int lo = dpp_row_sum(v & 0xffff), hi = dpp_row_sum(v >> 16);
    return (long long)hi * 65536ll + (long long)lo;
}
__device__ __forceinline__ unsigned dpp_from_next_lane(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);  // row_shl:1 -> lane i reads lane i+1
}
// acc += dot2(a of lane i+1, b): the second link of a bilinear chain takes the packed pair of the row below straight from the neighbour lane
__device__ __forceinline__ int dot2c_next_lane(int acc, unsigned a, unsigned b)
{
    asm("v_dot2c_i32_i16_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
// packed byte pair (c, c+1) of a row held as aligned words a[0..]
__device__ __forceinline__ unsigned row_pair(const unsigned* a, int c)
{
    const int w = c >> 2, q = c & 3;
    return q == 0 ? __builtin_amdgcn_perm(0u, a[w], 0x0c010c00u)
         : q == 1 ? __builtin_amdgcn_perm(0u, a[w], 0x0c020c01u)
         : q == 2 ? __builtin_amdgcn_perm(0u, a[w], 0x0c030c02u)
                  : __builtin_amdgcn_perm(a[w + 1], a[w], 0x0c040c03u);
}
// bilinear x32 samples of row r's NS strips at (x0, y0 + r) of image im: lane r holds row y0 + r, the row below comes from lane r + 1
template <int NS>
__device__ __forceinline__ void lkq_sample_row(const ImgDesc& im, int x0, int y0, int r, bool fast, unsigned wt, unsigned wb, unsigned* p01, unsigned* p23)
{
    unsigned top[NS + 1];
    load_row_words<NS + 1>(im, x0, y0 + r, fast, top);
    int v[4 * NS];
#pragma unroll
    for (int c = 0; c < 4 * NS; c++) {
        const unsigned pr = row_pair(top, c);
        v[c] = (dot2c_next_lane(dot2_first(pr, wt), pr, wb) << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
    }
#pragma unroll
    for (int j = 0; j < NS; j++) {
        p01[j] = pack_hi16(v[4 * j], v[4 * j + 1]);
        p23[j] = pack_hi16(v[4 * j + 2], v[4 * j + 3]);
    }
}

// NW5 = number of aligned 4-byte words a lane needs from one image row, starting at pixel (gx, gy): word i = bytes 4i..4i+3
template <int NWORDS>
__device__ __forceinline__ void load_row_words(const ImgDesc& im, int gx, int gy, bool fast, unsigned* a)
{
    if (fast) {
        const uintptr_t p = reinterpret_cast<uintptr_t>(im.p + (ptrdiff_t)gy * im.stride + gx);
        const unsigned sh = (unsigned)(p & 3);
        gptr_u32 ap = (gptr_u32)(p - sh);
        unsigned d[NWORDS + 1];
#pragma unroll
        for (int i = 0; i <= NWORDS; i++) d[i] = ap[i];
#pragma unroll
        for (int i = 0; i < NWORDS; i++) a[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
    } else {
        const uint8_t* row = im.p + (size_t)vh_reflect101_near(gy, im.h) * im.stride;
        // all column indices first (branch free), then all byte loads: one memory round trip per row
        unsigned b[4 * NWORDS];
#pragma unroll
        for (int k = 0; k < 4 * NWORDS; k++) b[k] = row[vh_reflect101_near(gx + k, im.w)];
#pragma unroll
        for (int i = 0; i < NWORDS; i++) a[i] = b[4 * i] | (b[4 * i + 1] << 8) | (b[4 * i + 2] << 16) | (b[4 * i + 3] << 24);
    }
}

template <int WIN>
__device__ __forceinline__ void lkq_level(const ImgDesc I, const ImgDesc J, int level, int top_level, int max_count, double eps2, float p0x,
                                          float p0y, float& nxo, float& nyo, int& status, float& err, int r, int& n_iter, int& n_setup,
                                          bool want_err)
{
    constexpr int NS = (WIN + 3) >> 2;  // strips per row
    const float half = (float)(WIN - 1) * 0.5f;
    const float lscale = __uint_as_float((unsigned)(127 - level) << 23);  // 2^-level exactly = (float)(1. / (1 << level)), without the f64 division
    float px = __fmul_rn(p0x, lscale), py = __fmul_rn(p0y, lscale);
    float nx, ny;
    if (level == top_level) { nx = px; ny = py; }
    else { nx = __fmul_rn(nxo, 2.f); ny = __fmul_rn(nyo, 2.f); }
    nxo = nx; nyo = ny;

    px = __fsub_rn(px, half); py = __fsub_rn(py, half);
    const int ipx = vh_floor(px), ipy = vh_floor(py);
    if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
        if (level == 0) { status = 0; err = 0.f; }
        return;
    }
    const Win w0 = bilinear_weights(__fsub_rn(px, (float)ipx), __fsub_rn(py, (float)ipy));
    n_setup++;

    int a11 = 0, a12 = 0, a22 = 0;
    // template gradients of this lane's row (registers); the template SAMPLES are not kept: sum (J - I) Ix = sum J Ix - cI (see lk3_level)
    uint2 tX[NS], tY[NS];
    int cI[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < NS; j++) { tX[j] = make_uint2(0, 0); tY[j] = make_uint2(0, 0); }
    // interior window: the (WIN+3)^2 patch [ipx-1, ipx+WIN+1] x [ipy-1, ipy+WIN+1] lies inside the level, so the V identity holds.  The aligned
    // 24-byte row reads may run a few bytes past either end of a row: harmless inside the level, excluded where that would leave it (first row to
    // the left, last row to the right)
    const bool fast_I = ipx >= 1 && ipy >= 1 && ipx + WIN + 3 <= I.w && ipy + WIN + 2 <= I.h && !(ipy == 1 && ipx < 4) &&
                        !(ipy + WIN + 2 == I.h && ipx + WIN + 8 > I.w);
    const unsigned w0t = pack16(w0.w00, w0.w01), w0b = pack16(w0.w10, w0.w11);
    if (fast_I) {
        // Interior window, all 16 lanes of the track (lane WIN feeds lane WIN-1).  Lane r reads patch rows r .. r+2 and builds V rows r and r+1 over
        // the NC = 4 NS + 2 columns its NS adjacent strips share (V = bilinear weights . patch, no rounding: see strip_setup_linear); V row r+2 is
        // the neighbour's second row (DPP).  Scharr of V, vertical pass first:  S = 3 (V_r + V_r+2) + 10 V_r+1,  dV = V_r+2 - V_r,
        //   Ix = S[c+2] - S[c],   Iy = 3 (dV[c] + dV[c+2]) + 10 dV[c+1]
        // both kept x4 with the rounding folded in (column c of S carries 2^14 c), so descale + int16 packing is the upper half (pack_hi16).
        constexpr int NC = 4 * NS + 2;
        unsigned a[3][NS + 1];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) load_row_words<NS + 1>(I, ipx - 1, ipy + r - 1 + rr, true, a[rr]);
        int S4[NC], dV[NC], V1[NC];
        auto column = [&](int c) {
            const unsigned q0 = row_pair(a[0], c), q1 = row_pair(a[1], c), q2 = row_pair(a[2], c);
            const int v0 = dot2(q1, w0b, dot2_first(q0, w0t));
            V1[c] = dot2(q2, w0b, dot2_first(q1, w0t));
            const int v2 = (int)dpp_from_next_lane((unsigned)V1[c]);
            S4[c] = mad24_v<40>(V1[c], mad24_s<12>(v0 + v2, c << W_BITS));
            dV[c] = v2 - v0;
        };
        column(0); column(1);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            // strip j needs columns 4j .. 4j+5: four new ones per strip, and a scheduling barrier per strip keeps the column window short
            // (left alone hipcc builds all 18 columns first: 157 VGPRs)
#pragma unroll
            for (int c = 0; c < 4; c++) column(4 * j + 2 + c);
            constexpr int full = 4;
            const int cnt = WIN - 4 * j < full ? WIN - 4 * j : full;
            int iv[4], ix[4], iy[4];  // the wanted int16 in the upper half of each
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int col = 4 * j + c;
                iv[c] = (V1[col + 1] << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
                ix[c] = S4[col + 2] - S4[col];
                iy[c] = mad24_v<40>(dV[col + 1], mad24_s<12>(dV[col] + dV[col + 2], 4 << (W_BITS - 1)));
            }
            // masking by selector (see setup_from_hg): cnt is a compile-time constant here, only the row mask is per lane
            const unsigned sel01 = r < WIN ? (cnt >= 2 ? 0x07060302u : 0x0c0c0302u) : 0x0c0c0c0cu;
            const unsigned sel23 = r < WIN ? (cnt >= 4 ? 0x07060302u : (cnt == 3 ? 0x0c0c0302u : 0x0c0c0c0cu)) : 0x0c0c0c0cu;
            const uint2 vI = make_uint2(pack_hi16(iv[0], iv[1]), pack_hi16(iv[2], iv[3]));
            const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ix[1], (unsigned)ix[0], sel01), __builtin_amdgcn_perm((unsigned)ix[3], (unsigned)ix[2], sel23));
            const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iy[1], (unsigned)iy[0], sel01), __builtin_amdgcn_perm((unsigned)iy[3], (unsigned)iy[2], sel23));
            tX[j] = vX; tY[j] = vY;
            a11 = dot2(vX.y, vX.y, dot2(vX.x, vX.x, a11));
            a12 = dot2(vX.y, vY.y, dot2(vX.x, vY.x, a12));
            a22 = dot2(vY.y, vY.y, dot2(vY.x, vY.x, a22));
            cI[0] = dot2(vI.y, vX.y, dot2(vI.x, vX.x, cI[0]));
            cI[1] = dot2(vI.y, vY.y, dot2(vI.x, vY.x, cI[1]));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // Window at the image border (most tracks on the tiny top levels of a 5-level pyramid).  The derivative image is 0 OUTSIDE the level, so the V
        // identity does not hold: the Scharr gradients of the integer pixels are built, zeroed outside, then interpolated like any other image.
        // Everything stays packed int16 (S = 3 (P_up + P_down) + 10 P <= 4080, D = P_down - P_up, Gx = S[k+1] - S[k-1], Gy = 3 (D[k-1] + D[k+1]) + 10 D[k]):
        // lane r holds pixel rows r-1 .. r+1 and gradient row r of the window as byte / int16 PAIRS (k, k+1), so a sample is two v_dot2 -- the
        // second one reading the pair of gradient row r+1 straight from lane r+1 (DPP) -- exactly like the search-image sampling.  All 16 lanes
        // take part (lane WIN feeds lane WIN-1).  REFLECT_101 is in the row loads; same integers as strip_setup<false>.
        unsigned a[3][NS + 1];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) load_row_words<NS + 1>(I, ipx - 1, ipy + r - 1 + rr, I.pad >= VH_LV_PAD, a[rr]);  // bordered level: plain dword loads
        // bit c of M: gradient pixel (ipx + c, ipy + r) lies inside the level
        const int ay = ipy + r, cs = max(0, -ipx), ce = min(4 * NS, I.w - 1 - ipx);
        const unsigned M = (ay >= 0 && ay < I.h && ce >= cs) ? (((2u << ce) - 1u) & ~((1u << cs) - 1u)) : 0u;
        short2v Sp[4 * NS + 2], Dp[4 * NS + 2];
        unsigned Pm[4 * NS + 2];
        auto colb = [&](int k) {
            const unsigned pt = row_pair(a[0], k), pb = row_pair(a[2], k);
            Pm[k] = row_pair(a[1], k);
            const short2v k3 = {3, 3}, k10 = {10, 10};
            Sp[k] = (as_s2(pt) + as_s2(pb)) * k3 + as_s2(Pm[k]) * k10;
            Dp[k] = as_s2(pb) - as_s2(pt);
        };
        colb(0); colb(1);
#pragma unroll
        for (int j = 0; j < NS; j++) {
#pragma unroll
            for (int c = 0; c < 4; c++) colb(4 * j + 2 + c);
            const int cnt = WIN - 4 * j < 4 ? WIN - 4 * j : 4;
            int iv[4], ix[4], iy[4];  // the wanted int16 in the upper half of each
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int col = 4 * j + c;
                const short2v k3 = {3, 3}, k10 = {10, 10};
                // (pixel col, col + 1) of the gradient row: Gx = S[k+1] - S[k-1] with k = col + 1 in patch columns
                const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)M, col, 1) & 0xffffu | ((unsigned)__builtin_amdgcn_sbfe((int)M, col + 1, 1) << 16);
                const unsigned gx = __builtin_bit_cast(unsigned, Sp[col + 2] - Sp[col]) & m;
                const unsigned gy = __builtin_bit_cast(unsigned, (Dp[col] + Dp[col + 2]) * k3 + Dp[col + 1] * k10) & m;
                ix[c] = (dot2c_next_lane(dot2_first(gx, w0t), gx, w0b) << 2) + (4 << (W_BITS - 1));
                iy[c] = (dot2c_next_lane(dot2_first(gy, w0t), gy, w0b) << 2) + (4 << (W_BITS - 1));
                iv[c] = (dot2c_next_lane(dot2_first(Pm[col + 1], w0t), Pm[col + 1], w0b) << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
            }
            const unsigned sel01 = r < WIN ? (cnt >= 2 ? 0x07060302u : 0x0c0c0302u) : 0x0c0c0c0cu;
            const unsigned sel23 = r < WIN ? (cnt >= 4 ? 0x07060302u : (cnt == 3 ? 0x0c0c0302u : 0x0c0c0c0cu)) : 0x0c0c0c0cu;
            const uint2 vI = make_uint2(pack_hi16(iv[0], iv[1]), pack_hi16(iv[2], iv[3]));
            const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ix[1], (unsigned)ix[0], sel01), __builtin_amdgcn_perm((unsigned)ix[3], (unsigned)ix[2], sel23));
            const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iy[1], (unsigned)iy[0], sel01), __builtin_amdgcn_perm((unsigned)iy[3], (unsigned)iy[2], sel23));
            tX[j] = vX; tY[j] = vY;
            a11 = dot2(vX.y, vX.y, dot2(vX.x, vX.x, a11));
            a12 = dot2(vX.y, vY.y, dot2(vX.x, vY.x, a12));
            a22 = dot2(vY.y, vY.y, dot2(vY.x, vY.x, a22));
            cI[0] = dot2(vI.y, vX.y, dot2(vI.x, vX.x, cI[0]));
            cI[1] = dot2(vI.y, vY.y, dot2(vI.x, vY.x, cI[1]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float A11 = __fmul_rn(i64_to_f32(row16_sum_wide(a11)), LK_FLT_SCALE), A12 = __fmul_rn(i64_to_f32(row16_sum_wide(a12)), LK_FLT_SCALE),
                A22 = __fmul_rn(i64_to_f32(row16_sum_wide(a22)), LK_FLT_SCALE);
    float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
    const float dA = __fsub_rn(A11, A22);
    const float disc = __fadd_rn(__fmul_rn(dA, dA), __fmul_rn(__fmul_rn(4.f, A12), A12));
    const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), vh_sqrtf(disc)), (float)(2 * WIN * WIN));
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
        if (level == 0) status = 0;
        return;
    }
    D = __fdiv_rn(1.f, D);

    nx = __fsub_rn(nx, half); ny = __fsub_rn(ny, half);
    float pdx = 0.f, pdy = 0.f;
    for (int it = 0; it < max_count; it++) {
        const int inx = vh_floor(nx), iny = vh_floor(ny);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(nx, (float)inx), __fsub_rn(ny, (float)iny)));
        const bool fast = J.pad >= VH_LV_PAD || (inx >= 0 && iny >= 0 && inx + WIN + 2 <= J.w && iny + WIN + 1 <= J.h && !(iny == 0 && inx < 3) &&
                                                 !(iny + WIN + 1 == J.h && inx + WIN + 9 > J.w));  // bordered level, or rows inside the level (exclusions as fast_I)
        n_iter++;
        unsigned p01[NS], p23[NS];
        lkq_sample_row<NS>(J, inx, iny, r, fast, w.wt, w.wb, p01, p23);  // lane r = row r (lane WIN holds the last bottom row)
        int b1 = -cI[0], b2 = -cI[1];
#pragma unroll
        for (int j = 0; j < NS; j++) {
            // rows >= WIN and samples beyond the window edge carry Ix = Iy = 0, so they add nothing
            b1 = dot2(p23[j], tX[j].y, dot2(p01[j], tX[j].x, b1));
            b2 = dot2(p23[j], tY[j].y, dot2(p01[j], tY[j].x, b2));
        }
        const float fb1 = __fmul_rn(i64_to_f32(row16_sum_wide(b1)), LK_FLT_SCALE), fb2 = __fmul_rn(i64_to_f32(row16_sum_wide(b2)), LK_FLT_SCALE);
        const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
        const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
        nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
        nxo = __fadd_rn(nx, half); nyo = __fadd_rn(ny, half);
        if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= eps2) break;
        if (it > 0 && fabsf(__fadd_rn(dx, pdx)) < 0.01f && fabsf(__fadd_rn(dy, pdy)) < 0.01f) {
            nxo = __fsub_rn(nxo, __fmul_rn(dx, 0.5f));
            nyo = __fsub_rn(nyo, __fmul_rn(dy, 0.5f));
            break;
        }
        pdx = dx; pdy = dy;
    }

    if (status && level == 0) {
        const float fx = __fsub_rn(nxo, half), fy = __fsub_rn(nyo, half);
        const int inx = vh_floor(fx), iny = vh_floor(fy);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) { status = 0; return; }
        if (!want_err) return;
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(fx, (float)inx), __fsub_rn(fy, (float)iny)));
        const bool fast = J.pad >= VH_LV_PAD || (inx >= 0 && iny >= 0 && inx + WIN + 2 <= J.w && iny + WIN + 1 <= J.h && !(iny == 0 && inx < 3) &&
                                                 !(iny + WIN + 1 == J.h && inx + WIN + 9 > J.w));  // bordered level, or rows inside the level (exclusions as fast_I)
        unsigned p01[NS], p23[NS], i01[NS], i23[NS];
        lkq_sample_row<NS>(J, inx, iny, r, fast, w.wt, w.wb, p01, p23);
        // the template samples again (not kept by the set-up): the same bilinear sampling of I at the template origin
        lkq_sample_row<NS>(I, ipx, ipy, r, fast_I || I.pad >= VH_LV_PAD, w0t, w0b, i01, i23);
        int se = 0;
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const short2v d01 = as_s2(p01[j]) - as_s2(i01[j]), d23 = as_s2(p23[j]) - as_s2(i23[j]);
            const int d[4] = {d01.x, d01.y, d23.x, d23.y};
            const int cnt = WIN - 4 * j < 4 ? WIN - 4 * j : 4;
#pragma unroll
            for (int c = 0; c < 4; c++) se += (c < cnt && r < WIN) ? (d[c] < 0 ? -d[c] : d[c]) : 0;
        }
        err = __fmul_rn(i64_to_f32(row16_sum_wide(se)), __fdiv_rn(1.f, (float)(32 * WIN * WIN)));
    }
}

template <int WIN>
__device__ __forceinline__ void lkq_track(const PyrDesc& PI, const PyrDesc& PJ, int max_count, double eps2, float px, float py, float& ox,
                                          float& oy, int& status, float& err, int r, int& n_iter, int& n_setup, bool want_err)
{
    const int nl = min(PI.nlevels, PJ.nlevels);
    status = 1;
    err = 0.f;
    ox = 0.f; oy = 0.f;
    for (int level = nl - 1; level >= 0; level--)
        lkq_level<WIN>(PI.lv[level], PJ.lv[level], level, nl - 1, max_count, eps2, px, py, ox, oy, status, err, r, n_iter, n_setup, want_err);
}

template <int WIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_lk_q(const void* job_tab, size_t tab_stride, unsigned grp)
{
    static_assert(WIN <= 15, "lane WIN of every 16-lane row carries the extra bottom row");
    unsigned blk_x, blk_y;
    lk_block_xy<true>(blk_x, blk_y, grp);
    const LKJob& job = *reinterpret_cast<const LKJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blk_y * tab_stride);
    const int n = LK_N_OF(job);
    const int slot = (int)blk_x * 4 + (threadIdx.x >> 4);
    if (slot >= n) return;
    const int pt = job.order ? ((gptr_i32)job.order)[slot] : slot;  // launch slot -> point (LKJob::order)  // whole 16-lane rows leave together
    const int r = threadIdx.x & 15;

    const float qx = ((gptr_f32)job.p_in)[2 * pt], qy = ((gptr_f32)job.p_in)[2 * pt + 1];
    const float px = __fsub_rn(__fmul_rn(qx, job.in_scale), job.in_off[0]);
    const float py = __fsub_rn(__fmul_rn(qy, job.in_scale), job.in_off[1]);

    float fx, fy, err;
    int st, n_iter = 0, n_setup = 0;
    lkq_track<WIN>(job.I, job.J, job.max_count, job.eps2, px, py, fx, fy, st, err, r, n_iter, n_setup, job.err_out != nullptr);
    float fbe = 0.f;
    if (job.fbt >= 0.f) {
        float bx = 0.f, by = 0.f, e2;
        int st2 = 0;
        if (st || job.fbe_out) lkq_track<WIN>(job.J, job.I, job.max_count, job.eps2, fx, fy, bx, by, st2, e2, r, n_iter, n_setup, false);
        const float ddx = __fsub_rn(px, bx), ddy = __fsub_rn(py, by);
        fbe = vh_sqrtf(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)));
        st = st && st2 && (fbe < job.fbt);
    }
    if (r == 0) {
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
            unsigned long long* st = job.stats + (size_t)((unsigned)slot & (VH_LK_STAT_SLOTS - 1)) * 16;  // (counters spread over VH_LK_STAT_SLOTS lines: see StreamWS::lk_stats)
            atomicAdd(&st[0], (unsigned long long)n_iter);
            atomicAdd(&st[1], (unsigned long long)n_setup);
        }
    }
}

// =================================================================================================================
// v5: eighth-wave kernel for the 15x15 coarse stages at full load.
// 8 tracks per wavefront: a track owns 8 consecutive lanes (half a DPP row), lane r owns window rows 2r and 2r+1 (row 15 of lane 7 is a
// dummy that carries zero gradients).  What k_lk_q spends per 4 tracks on the per-track "uniform" work -- weights, the three window sums and
// minEig, the 2x2 solve, the stop rules, every DPP reduction: ~37 % of its instructions -- serves 8 tracks here, and the template set-up
// shares V rows inside a lane: a lane builds V rows 2r, 2r+1, 2r+2 from its four patch rows (3 per 2 window rows instead of 2 per row) and
// takes V row 2r+3 from lane r+1.  In the Newton iterations a lane loads its own two search rows; the row below its second row is the first
// row of lane r+1 (v_dot2c_i32_i16_dpp).  Window sums are 8-lane DPP reductions (quad_perm, quad_perm, row_half_mirror).  Lane 7 of a track
// reads lane 0 of the next track through DPP only for its dummy row.  Same integers, same float sequence per track: bit-identical results.
// =================================================================================================================
__device__ __forceinline__ int dpp_oct_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror: lane i <-> lane 7 - i of its half row
    return v;
}
// exact 8-lane sum of int32 partials as the float32 the int64 -> float32 conversion gives: hi * 2^16 + lo is exact in float64 (|hi| < 2^19, lo < 2^19),
// the final conversion rounds once
__device__ __forceinline__ float oct_sum_f32(int v)
{
    const int lo = dpp_oct_sum(v & 0xffff), hi = dpp_oct_sum(v >> 16);
    return (float)__fma_rn((double)hi, 65536.0, (double)lo);
}
// bilinear x32 samples of window rows 2r and 2r+1 (NS strips each) at (x0, y0) of image im: the lane holds image rows y0 + 2r and y0 + 2r + 1,
// the row below them is the first row of lane r + 1
template <int NS, int WIN>
__device__ __forceinline__ void lko_sample_rows(const ImgDesc& im, int x0, int y0, int r, bool fast, unsigned wt, unsigned wb, unsigned* a01,
                                                unsigned* a23, unsigned* b01, unsigned* b23)
{
    unsigned top[NS + 1], mid[NS + 1];
    load_row_words<NS + 1>(im, x0, y0 + 2 * r, fast, top);
    load_row_words<NS + 1>(im, x0, y0 + 2 * r + 1, fast, mid);
    constexpr int SH = 16 - (W_BITS - 5), RND = 1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5));
#pragma unroll
    for (int j = 0; j < NS; j++) {
        int va[4] = {0, 0, 0, 0}, vb[4] = {0, 0, 0, 0};  // columns >= WIN are never used (zero template gradients there): not computed
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (4 * j + c >= WIN) continue;
            const unsigned pa = row_pair(top, 4 * j + c), pb = row_pair(mid, 4 * j + c);
            va[c] = (dot2(pb, wb, dot2_first(pa, wt)) << SH) + RND;
            vb[c] = (dot2c_next_lane(dot2_first(pb, wt), pa, wb) << SH) + RND;
        }
        a01[j] = pack_hi16(va[0], va[1]); a23[j] = pack_hi16(va[2], va[3]);
        b01[j] = pack_hi16(vb[0], vb[1]); b23[j] = pack_hi16(vb[2], vb[3]);
    }
}

template <int WIN>
__device__ __forceinline__ void lko_level(const ImgDesc I, const ImgDesc J, int level, int top_level, int max_count, double eps2, float p0x,
                                          float p0y, float& nxo, float& nyo, int& status, float& err, int r, int& n_iter, int& n_setup,
                                          bool want_err)
{
    constexpr int NS = (WIN + 3) >> 2;  // strips per row
    const float half = (float)(WIN - 1) * 0.5f;
    const float lscale = __uint_as_float((unsigned)(127 - level) << 23);
    float px = __fmul_rn(p0x, lscale), py = __fmul_rn(p0y, lscale);
    float nx, ny;
    if (level == top_level) { nx = px; ny = py; }
    else { nx = __fmul_rn(nxo, 2.f); ny = __fmul_rn(nyo, 2.f); }
    nxo = nx; nyo = ny;

    px = __fsub_rn(px, half); py = __fsub_rn(py, half);
    const int ipx = vh_floor(px), ipy = vh_floor(py);
    if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
        if (level == 0) { status = 0; err = 0.f; }
        return;
    }
    const Win w0 = bilinear_weights(__fsub_rn(px, (float)ipx), __fsub_rn(py, (float)ipy));
    n_setup++;

    int a11 = 0, a12 = 0, a22 = 0;
    uint2 tXa[NS], tYa[NS], tXb[NS], tYb[NS];  // template gradients of window rows 2r (a) and 2r + 1 (b)
    int cI[2] = {0, 0};
    const bool fast_I = ipx >= 1 && ipy >= 1 && ipx + WIN + 3 <= I.w && ipy + WIN + 2 <= I.h && !(ipy == 1 && ipx < 4) &&
                        !(ipy + WIN + 2 == I.h && ipx + WIN + 8 > I.w);
    const unsigned w0t = pack16(w0.w00, w0.w01), w0b = pack16(w0.w10, w0.w11);
    const bool rowb = 2 * r + 1 < WIN;  // (row 2r is always a window row)
    // accumulate one strip of one window row: template samples vI, gradients vX / vY (already masked)
    auto accumulate = [&](uint2 vI, uint2 vX, uint2 vY) {
        a11 = dot2(vX.y, vX.y, dot2(vX.x, vX.x, a11));
        a12 = dot2(vX.y, vY.y, dot2(vX.x, vY.x, a12));
        a22 = dot2(vY.y, vY.y, dot2(vY.x, vY.x, a22));
        cI[0] = dot2(vI.y, vX.y, dot2(vI.x, vX.x, cI[0]));
        cI[1] = dot2(vI.y, vY.y, dot2(vI.x, vY.x, cI[1]));
    };
    if (fast_I) {
        // Lane r reads patch rows 2r .. 2r+3 (patch row 0 = image row ipy - 1) and builds V rows 2r, 2r+1, 2r+2 over the NC columns its strips share
        // (V = bilinear weights . patch, no rounding); V row 2r+3 is the neighbour's V row 2(r+1)+1 (DPP).  Window row w uses V rows w, w+1, w+2:
        //   S = 3 (V_w + V_w+2) + 10 V_w+1,  dV = V_w+2 - V_w,  Ix = S[c+2] - S[c],  Iy = 3 (dV[c] + dV[c+2]) + 10 dV[c+1]   (see lkq_level)
        constexpr int NC = 4 * NS + 2;
        unsigned a[4][NS + 1];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) load_row_words<NS + 1>(I, ipx - 1, ipy - 1 + 2 * r + rr, true, a[rr]);
        int SA[NC], SB[NC], dA[NC], dB[NC], CA[NC], CB[NC];
        auto column = [&](int c) {
            const unsigned q0 = row_pair(a[0], c), q1 = row_pair(a[1], c), q2 = row_pair(a[2], c), q3 = row_pair(a[3], c);
            const int va = dot2(q1, w0b, dot2_first(q0, w0t));
            const int vb = dot2(q2, w0b, dot2_first(q1, w0t));
            const int vc = dot2(q3, w0b, dot2_first(q2, w0t));
            const int vd = (int)dpp_from_next_lane((unsigned)vb);
            CA[c] = vb; CB[c] = vc;
            SA[c] = mad24_v<40>(vb, mad24_s<12>(va + vc, c << W_BITS));
            dA[c] = vc - va;
            SB[c] = mad24_v<40>(vc, mad24_s<12>(vb + vd, c << W_BITS));
            dB[c] = vd - vb;
        };
        column(0); column(1);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int cnt = WIN - 4 * j < 4 ? WIN - 4 * j : 4;
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (c < cnt) column(4 * j + 2 + c);  // (window column k uses V columns k .. k+2)
            const unsigned s01 = cnt >= 2 ? 0x07060302u : 0x0c0c0302u, s23 = cnt >= 4 ? 0x07060302u : (cnt == 3 ? 0x0c0c0302u : 0x0c0c0c0cu);
            const unsigned sb01 = rowb ? s01 : 0x0c0c0c0cu, sb23 = rowb ? s23 : 0x0c0c0c0cu;
            int iv[4] = {0, 0, 0, 0}, ix[4] = {0, 0, 0, 0}, iy[4] = {0, 0, 0, 0};
            // window row 2r
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c >= cnt) continue;
                const int col = 4 * j + c;
                iv[c] = (CA[col + 1] << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
                ix[c] = SA[col + 2] - SA[col];
                iy[c] = mad24_v<40>(dA[col + 1], mad24_s<12>(dA[col] + dA[col + 2], 4 << (W_BITS - 1)));
            }
            {
                const uint2 vI = make_uint2(pack_hi16(iv[0], iv[1]), pack_hi16(iv[2], iv[3]));
                const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ix[1], (unsigned)ix[0], s01), __builtin_amdgcn_perm((unsigned)ix[3], (unsigned)ix[2], s23));
                const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iy[1], (unsigned)iy[0], s01), __builtin_amdgcn_perm((unsigned)iy[3], (unsigned)iy[2], s23));
                tXa[j] = vX; tYa[j] = vY;
                accumulate(vI, vX, vY);
            }
            // window row 2r + 1
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c >= cnt) continue;
                const int col = 4 * j + c;
                iv[c] = (CB[col + 1] << (16 - (W_BITS - 5))) + (1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5)));
                ix[c] = SB[col + 2] - SB[col];
                iy[c] = mad24_v<40>(dB[col + 1], mad24_s<12>(dB[col] + dB[col + 2], 4 << (W_BITS - 1)));
            }
            {
                const uint2 vI = make_uint2(pack_hi16(iv[0], iv[1]), pack_hi16(iv[2], iv[3]));
                const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ix[1], (unsigned)ix[0], sb01), __builtin_amdgcn_perm((unsigned)ix[3], (unsigned)ix[2], sb23));
                const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iy[1], (unsigned)iy[0], sb01), __builtin_amdgcn_perm((unsigned)iy[3], (unsigned)iy[2], sb23));
                tXb[j] = vX; tYb[j] = vY;
                accumulate(vI, vX, vY);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // Window at the image border: the derivative image is 0 OUTSIDE the level, so the Scharr gradients of the integer pixels are built (packed
        // int16 pairs (k, k+1), see lkq_level), zeroed outside, then interpolated like any other image.  Lane r holds pixel rows 2r-1 .. 2r+2 of the
        // window (image rows ipy + 2r - 1 ..) and gradient rows 2r, 2r+1; gradient / pixel row 2r+2 is lane r+1's first row (DPP).
        unsigned a[4][NS + 1];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) load_row_words<NS + 1>(I, ipx - 1, ipy - 1 + 2 * r + rr, I.pad >= VH_LV_PAD, a[rr]);
        const int ay = ipy + 2 * r, cs = max(0, -ipx), ce = min(4 * NS, I.w - 1 - ipx);
        const unsigned Mc = ce >= cs ? (((2u << ce) - 1u) & ~((1u << cs) - 1u)) : 0u;
        const unsigned MA = (ay >= 0 && ay < I.h) ? Mc : 0u, MB = (ay + 1 >= 0 && ay + 1 < I.h) ? Mc : 0u;
        short2v SpA[4 * NS + 2], DpA[4 * NS + 2], SpB[4 * NS + 2], DpB[4 * NS + 2];
        unsigned PA[4 * NS + 2], PB[4 * NS + 2];
        const short2v k3 = {3, 3}, k10 = {10, 10};
        auto colb = [&](int k) {
            const unsigned p0 = row_pair(a[0], k), p3 = row_pair(a[3], k);
            PA[k] = row_pair(a[1], k);
            PB[k] = row_pair(a[2], k);
            SpA[k] = (as_s2(p0) + as_s2(PB[k])) * k3 + as_s2(PA[k]) * k10;
            DpA[k] = as_s2(PB[k]) - as_s2(p0);
            SpB[k] = (as_s2(PA[k]) + as_s2(p3)) * k3 + as_s2(PB[k]) * k10;
            DpB[k] = as_s2(p3) - as_s2(PA[k]);
        };
        colb(0); colb(1);
        constexpr int SH = 16 - (W_BITS - 5), RND = 1 << (W_BITS - 5 - 1 + 16 - (W_BITS - 5));
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int cnt = WIN - 4 * j < 4 ? WIN - 4 * j : 4;
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (c < cnt) colb(4 * j + 2 + c);
            const unsigned s01 = cnt >= 2 ? 0x07060302u : 0x0c0c0302u, s23 = cnt >= 4 ? 0x07060302u : (cnt == 3 ? 0x0c0c0302u : 0x0c0c0c0cu);
            const unsigned sb01 = rowb ? s01 : 0x0c0c0c0cu, sb23 = rowb ? s23 : 0x0c0c0c0cu;
            int ivA[4] = {0, 0, 0, 0}, ixA[4] = {0, 0, 0, 0}, iyA[4] = {0, 0, 0, 0}, ivB[4] = {0, 0, 0, 0}, ixB[4] = {0, 0, 0, 0}, iyB[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c >= cnt) continue;
                const int col = 4 * j + c;
                const unsigned ma = (unsigned)__builtin_amdgcn_sbfe((int)MA, col, 1) & 0xffffu | ((unsigned)__builtin_amdgcn_sbfe((int)MA, col + 1, 1) << 16);
                const unsigned mb = (unsigned)__builtin_amdgcn_sbfe((int)MB, col, 1) & 0xffffu | ((unsigned)__builtin_amdgcn_sbfe((int)MB, col + 1, 1) << 16);
                const unsigned gxa = __builtin_bit_cast(unsigned, SpA[col + 2] - SpA[col]) & ma;
                const unsigned gya = __builtin_bit_cast(unsigned, (DpA[col] + DpA[col + 2]) * k3 + DpA[col + 1] * k10) & ma;
                const unsigned gxb = __builtin_bit_cast(unsigned, SpB[col + 2] - SpB[col]) & mb;
                const unsigned gyb = __builtin_bit_cast(unsigned, (DpB[col] + DpB[col + 2]) * k3 + DpB[col + 1] * k10) & mb;
                ixA[c] = (dot2(gxb, w0b, dot2_first(gxa, w0t)) << 2) + (4 << (W_BITS - 1));
                iyA[c] = (dot2(gyb, w0b, dot2_first(gya, w0t)) << 2) + (4 << (W_BITS - 1));
                ivA[c] = (dot2(PB[col + 1], w0b, dot2_first(PA[col + 1], w0t)) << SH) + RND;
                ixB[c] = (dot2c_next_lane(dot2_first(gxb, w0t), gxa, w0b) << 2) + (4 << (W_BITS - 1));
                iyB[c] = (dot2c_next_lane(dot2_first(gyb, w0t), gya, w0b) << 2) + (4 << (W_BITS - 1));
                ivB[c] = (dot2c_next_lane(dot2_first(PB[col + 1], w0t), PA[col + 1], w0b) << SH) + RND;
            }
            {
                const uint2 vI = make_uint2(pack_hi16(ivA[0], ivA[1]), pack_hi16(ivA[2], ivA[3]));
                const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ixA[1], (unsigned)ixA[0], s01), __builtin_amdgcn_perm((unsigned)ixA[3], (unsigned)ixA[2], s23));
                const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iyA[1], (unsigned)iyA[0], s01), __builtin_amdgcn_perm((unsigned)iyA[3], (unsigned)iyA[2], s23));
                tXa[j] = vX; tYa[j] = vY;
                accumulate(vI, vX, vY);
            }
            {
                const uint2 vI = make_uint2(pack_hi16(ivB[0], ivB[1]), pack_hi16(ivB[2], ivB[3]));
                const uint2 vX = make_uint2(__builtin_amdgcn_perm((unsigned)ixB[1], (unsigned)ixB[0], sb01), __builtin_amdgcn_perm((unsigned)ixB[3], (unsigned)ixB[2], sb23));
                const uint2 vY = make_uint2(__builtin_amdgcn_perm((unsigned)iyB[1], (unsigned)iyB[0], sb01), __builtin_amdgcn_perm((unsigned)iyB[3], (unsigned)iyB[2], sb23));
                tXb[j] = vX; tYb[j] = vY;
                accumulate(vI, vX, vY);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float A11 = __fmul_rn(oct_sum_f32(a11), LK_FLT_SCALE), A12 = __fmul_rn(oct_sum_f32(a12), LK_FLT_SCALE),
                A22 = __fmul_rn(oct_sum_f32(a22), LK_FLT_SCALE);
    float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
    const float dAf = __fsub_rn(A11, A22);
    const float disc = __fadd_rn(__fmul_rn(dAf, dAf), __fmul_rn(__fmul_rn(4.f, A12), A12));
    const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), vh_sqrtf(disc)), (float)(2 * WIN * WIN));
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
        if (level == 0) status = 0;
        return;
    }
    D = __fdiv_rn(1.f, D);

    nx = __fsub_rn(nx, half); ny = __fsub_rn(ny, half);
    float pdx = 0.f, pdy = 0.f;
    for (int it = 0; it < max_count; it++) {
        const int inx = vh_floor(nx), iny = vh_floor(ny);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(nx, (float)inx), __fsub_rn(ny, (float)iny)));
        const bool fast = J.pad >= VH_LV_PAD || (inx >= 0 && iny >= 0 && inx + WIN + 2 <= J.w && iny + WIN + 1 <= J.h && !(iny == 0 && inx < 3) &&
                                                 !(iny + WIN + 1 == J.h && inx + WIN + 9 > J.w));
        n_iter++;
        unsigned pa01[NS], pa23[NS], pb01[NS], pb23[NS];
        lko_sample_rows<NS, WIN>(J, inx, iny, r, fast, w.wt, w.wb, pa01, pa23, pb01, pb23);
        int b1 = -cI[0], b2 = -cI[1];
#pragma unroll
        for (int j = 0; j < NS; j++) {
            b1 = dot2(pa23[j], tXa[j].y, dot2(pa01[j], tXa[j].x, b1));
            b2 = dot2(pa23[j], tYa[j].y, dot2(pa01[j], tYa[j].x, b2));
            b1 = dot2(pb23[j], tXb[j].y, dot2(pb01[j], tXb[j].x, b1));
            b2 = dot2(pb23[j], tYb[j].y, dot2(pb01[j], tYb[j].x, b2));
        }
        const float fb1 = __fmul_rn(oct_sum_f32(b1), LK_FLT_SCALE), fb2 = __fmul_rn(oct_sum_f32(b2), LK_FLT_SCALE);
        const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
        const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
        nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
        nxo = __fadd_rn(nx, half); nyo = __fadd_rn(ny, half);
        if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= eps2) break;
        if (it > 0 && fabsf(__fadd_rn(dx, pdx)) < 0.01f && fabsf(__fadd_rn(dy, pdy)) < 0.01f) {
            nxo = __fsub_rn(nxo, __fmul_rn(dx, 0.5f));
            nyo = __fsub_rn(nyo, __fmul_rn(dy, 0.5f));
            break;
        }
        pdx = dx; pdy = dy;
    }

    if (status && level == 0) {
        const float fx = __fsub_rn(nxo, half), fy = __fsub_rn(nyo, half);
        const int inx = vh_floor(fx), iny = vh_floor(fy);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) { status = 0; return; }
        if (!want_err) return;
        const StripWeights w = strip_weights(bilinear_weights(__fsub_rn(fx, (float)inx), __fsub_rn(fy, (float)iny)));
        const bool fast = J.pad >= VH_LV_PAD || (inx >= 0 && iny >= 0 && inx + WIN + 2 <= J.w && iny + WIN + 1 <= J.h && !(iny == 0 && inx < 3) &&
                                                 !(iny + WIN + 1 == J.h && inx + WIN + 9 > J.w));
        unsigned pa01[NS], pa23[NS], pb01[NS], pb23[NS], ia01[NS], ia23[NS], ib01[NS], ib23[NS];
        lko_sample_rows<NS, WIN>(J, inx, iny, r, fast, w.wt, w.wb, pa01, pa23, pb01, pb23);
        lko_sample_rows<NS, WIN>(I, ipx, ipy, r, fast_I || I.pad >= VH_LV_PAD, w0t, w0b, ia01, ia23, ib01, ib23);
        int se = 0;
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const short2v da01 = as_s2(pa01[j]) - as_s2(ia01[j]), da23 = as_s2(pa23[j]) - as_s2(ia23[j]);
            const short2v db01 = as_s2(pb01[j]) - as_s2(ib01[j]), db23 = as_s2(pb23[j]) - as_s2(ib23[j]);
            const int da[4] = {da01.x, da01.y, da23.x, da23.y}, db[4] = {db01.x, db01.y, db23.x, db23.y};
            const int cnt = WIN - 4 * j < 4 ? WIN - 4 * j : 4;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                se += c < cnt ? (da[c] < 0 ? -da[c] : da[c]) : 0;
                se += (c < cnt && rowb) ? (db[c] < 0 ? -db[c] : db[c]) : 0;
            }
        }
        err = __fmul_rn(oct_sum_f32(se), __fdiv_rn(1.f, (float)(32 * WIN * WIN)));
    }
}

template <int WIN>
__device__ __forceinline__ void lko_track(const PyrDesc& PI, const PyrDesc& PJ, int max_count, double eps2, float px, float py, float& ox,
                                          float& oy, int& status, float& err, int r, int& n_iter, int& n_setup, bool want_err)
{
    const int nl = min(PI.nlevels, PJ.nlevels);
    status = 1;
    err = 0.f;
    ox = 0.f; oy = 0.f;
    for (int level = nl - 1; level >= 0; level--)
        lko_level<WIN>(PI.lv[level], PJ.lv[level], level, nl - 1, max_count, eps2, px, py, ox, oy, status, err, r, n_iter, n_setup, want_err);
}

template <int WIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_lk_o(const void* job_tab, size_t tab_stride, unsigned grp)
{
    static_assert(WIN == 15, "8 lanes x 2 rows: row 15 is the dummy that feeds nothing");
    unsigned blk_x, blk_y;
    lk_block_xy<true>(blk_x, blk_y, grp);
    const LKJob& job = *reinterpret_cast<const LKJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blk_y * tab_stride);
    const int n = LK_N_OF(job);
    const int slot = (int)blk_x * 8 + (threadIdx.x >> 3);
    if (slot >= n) return;  // whole 8-lane groups leave together
    const int pt = job.order ? ((gptr_i32)job.order)[slot] : slot;  // launch slot -> point (LKJob::order)
    const int r = threadIdx.x & 7;

    const float qx = ((gptr_f32)job.p_in)[2 * pt], qy = ((gptr_f32)job.p_in)[2 * pt + 1];
    const float px = __fsub_rn(__fmul_rn(qx, job.in_scale), job.in_off[0]);
    const float py = __fsub_rn(__fmul_rn(qy, job.in_scale), job.in_off[1]);

    float fx, fy, err;
    int st, n_iter = 0, n_setup = 0;
    lko_track<WIN>(job.I, job.J, job.max_count, job.eps2, px, py, fx, fy, st, err, r, n_iter, n_setup, job.err_out != nullptr);
    float fbe = 0.f;
    if (job.fbt >= 0.f) {
        float bx = 0.f, by = 0.f, e2;
        int st2 = 0;
        if (st || job.fbe_out) lko_track<WIN>(job.J, job.I, job.max_count, job.eps2, fx, fy, bx, by, st2, e2, r, n_iter, n_setup, false);
        const float ddx = __fsub_rn(px, bx), ddy = __fsub_rn(py, by);
        fbe = vh_sqrtf(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)));
        st = st && st2 && (fbe < job.fbt);
    }
    if (r == 0) {
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
            unsigned long long* st = job.stats + (size_t)((unsigned)slot & (VH_LK_STAT_SLOTS - 1)) * 16;  // (counters spread over VH_LK_STAT_SLOTS lines: see StreamWS::lk_stats)
            atomicAdd(&st[0], (unsigned long long)n_iter);
            atomicAdd(&st[1], (unsigned long long)n_setup);
        }
    }
}

template <int WIN>
static int launch_lko(const void* job_tab, size_t tab_stride, int batch, int max_n, hipStream_t s)
{
    static const unsigned grp = getenv("VH_LKO_G") ? (unsigned)atoi(getenv("VH_LKO_G")) : 64u;  // (environment: experiments only)
    hipLaunchKernelGGL(k_lk_o<WIN>, dim3((max_n + 7) / 8, batch), dim3(64), 0, s, job_tab, tab_stride, lk_group_arg(grp, batch));
    return 0;
}

template <int WIN>
static int launch_lkq(const void* job_tab, size_t tab_stride, int batch, int max_n, hipStream_t s)
{
    static const unsigned grp = getenv("VH_LKQ_G") ? (unsigned)atoi(getenv("VH_LKQ_G")) : 64u;  // (environment: experiments only)
    hipLaunchKernelGGL(k_lk_q<WIN>, dim3((max_n + 3) / 4, batch), dim3(64), 0, s, job_tab, tab_stride, lk_group_arg(grp, batch));
    return 0;
}

// test hook (vh_debug_lk3_tpw): launch slots per workgroup of the one-wavefront LDS-staged kernel -- 0 = chosen by load, n > 0 = n (clamped to MAX_TPW).
// PROCESS-WIDE atomic like the other hooks; every value gives bit-identical results (VH_LK3_TPW: experiments only, the initial value).
static std::atomic<int> g_lk3_tpw{getenv("VH_LK3_TPW") ? atoi(getenv("VH_LK3_TPW")) : 0};
void vh_lk3_set_tpw(int n) { g_lk3_tpw.store(n < 0 ? 0 : n, std::memory_order_relaxed); }

template <int WIN, int NW, int M>
static int launch_lk3(const void* job_tab, size_t tab_stride, int batch, int max_n, hipStream_t s, int* tpw_out)
{
    // VH_LK_LDS_PAD (experiments only): extra dynamic LDS per workgroup = fewer resident wavefronts per SIMD (the occupancy sensitivity behind DESIGN.md section 9)
    static const int pad = [] { const char* e = getenv("VH_LK_LDS_PAD"); return e ? atoi(e) : 0; }();
    const int lds = LK3<WIN, NW, M>::LDS_BYTES + pad;
    static const unsigned grp = getenv("VH_LK3_G") ? (unsigned)atoi(getenv("VH_LK3_G")) : 16u;  // (environment: experiments only)
    // Launch slots per workgroup (one-wavefront kernel): a 51 x 51 track keeps a workgroup for ~22 us, and starting one (dispatch, LDS allocation, the block
    // remap, the first descriptor loads) is not free: 4 consecutive slots per workgroup measured 3620 -> 3520 us per launch at 512 000 tracks (2: 3580,
    // 8: 3545; A/B on one box).  Only where the launch still has many workgroups per resident slot (256 CUs x 12): below that the tail would cost more.
    const int tpw_env = g_lk3_tpw.load(std::memory_order_relaxed);
    const long long tracks = (long long)max_n * batch;
    const int tpw_auto = NW != 1 ? 1 : tracks >= 98304 ? 4 : tracks >= 49152 ? 2 : 1;
    const unsigned tpw = (unsigned)std::min(NW == 1 && tpw_env > 0 ? tpw_env : tpw_auto, LK3<WIN, NW, M>::MAX_TPW);
    if (tpw_out) *tpw_out = (int)tpw;
    hipLaunchKernelGGL((k_lk3<WIN, NW, M>), dim3((max_n + tpw - 1) / tpw, batch), dim3(64 * NW), lds, s, job_tab, tab_stride, lk_group_arg(grp, batch), tpw);
    return 0;
}

template <int WIN_T>
static int launch_strip(const void* job_tab, size_t tab_stride, int batch, int max_n, int win, hipStream_t s)
{
    const int kmax = (((win + 3) >> 2) * win + 63) / 64;
    const size_t lds = (size_t)kmax * 64 * 24;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lk_strip<WIN_T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_lk_strip<WIN_T>, dim3(max_n, batch), dim3(64), lds, s, job_tab, tab_stride);
    return 0;
}

// test hook: 1 = per-sample kernel, 2 = strip kernel, 3 = LDS-staged kernel, 4 = 4-tracks-per-wave kernel (15x15), 5 / 6 / 7 = LDS-staged 51x51 kernel with
// 1 / 2 / 4 wavefronts per track, 8 = 8-tracks-per-wave kernel (15x15), 0 = default routing.
// PROCESS-WIDE (every context of the process is re-routed; an atomic, so a flip while another thread launches is a clean switch between two
// bit-identical implementations, never a torn value) -- a test / experiment switch, not a per-stream control.
static std::atomic<int> g_lk_force_generic{getenv("VH_LK_FORCE") ? atoi(getenv("VH_LK_FORCE")) : 0};  // (environment: experiments only)
void vh_lk_force_generic(int on) { g_lk_force_generic.store(on, std::memory_order_relaxed); }
static const long long g_lko_min_tracks = getenv("VH_LKO_MIN") ? atoll(getenv("VH_LKO_MIN")) : 10000;  // (environment: experiments only)

// Which kernel a launch of `batch` streams x `max_n` track slots with window `win` takes (the ids of the test hook above; 2 = the strip kernel of any
// window <= 63, 1 = the per-sample kernel).  vh_launch_lk follows exactly this decision, and the profiling records report it (vh_profile_lk_routes), so
// nobody has to mirror the thresholds.
//
// Default routing, re-measured late in round 5 -- the earlier thresholds had been measured while every LK launch carried ~45 us of serialised statistics
// atomics (StreamWS::lk_stats), which hid what the kernels themselves cost at a few thousand tracks.  51x51 fine stage: the LDS-staged kernel with ONE
// wavefront per track from 640 tracks in flight (2000 tracks: 28.8 us against 33.2 / 43.2 with 2 / 4 wavefronts per track; 768: 37.1 / 39.7 / 41.1 for
// the whole vh_pyr_lk call), 4 wavefronts per track below (128 tracks: 30.8 against 35.3); 2 per track is never the fastest and is kept for the tests.
// 15x15 coarse stages: the one-wave-per-track strip kernel up to 3000 tracks (2000: 20.2 / 27.9 us against 22.2 / 29.4 for 4 tracks per wavefront; 4000:
// 25.6 / 36.6 against 23.1 / 29.2), 4 tracks per wavefront up to 10 000, 8 per wavefront above (12 000: 33.6 / 50.2 against 35.4 / 55.9 us; 8000: a tie).
int vh_lk_route(int batch, int max_n, int win)
{
    const int force = g_lk_force_generic.load(std::memory_order_relaxed);
    const long long tracks = (long long)max_n * batch;
    const bool force_nw = force >= 5 && force <= 7;
    if (force == 0 || force == 3 || (force_nw && win == 51)) {
        if (win == 15 && force == 3) return 3;
        if (win == 51) {
            const int nw = force_nw ? (1 << (force - 5)) : (force == 3 ? 4 : tracks >= 640 ? 1 : 4);
            return nw == 1 ? 5 : nw == 2 ? 6 : 7;
        }
    }
    if (win == 15 && (force == 8 || ((force == 0 || force_nw) && tracks >= g_lko_min_tracks))) return 8;
    if (win == 15 && (force == 4 || ((force == 0 || force_nw) && tracks >= 3000))) return 4;
    if (force != 1 && win <= 63) return 2;  // (modes 5..7 with another window than 51: default routing); int32 per-lane partial sums are exact up to 16 strips per lane
    return 1;
}

const char* vh_lk_route_name(int route, int win)
{
    switch (route) {
    case 1: return "k_lk";
    case 2: return win == 15 ? "k_lk_strip<15>" : win == 51 ? "k_lk_strip<51>" : "k_lk_strip<0>";
    case 3: return "k_lk3<15, 1, 6>";
    case 4: return "k_lk_q<15>";
    case 5: return "k_lk3<51, 1, 4>";
    case 6: return "k_lk3<51, 2, 4>";
    case 7: return "k_lk3<51, 4, 4>";
    case 8: return "k_lk_o<15>";
    default: return "none";
    }
}

int vh_launch_lk(const void* job_tab, size_t tab_stride, int batch, int max_n, int win, hipStream_t s, int* route_out, int* tpw_out)
{
    if (route_out) *route_out = 0;
    if (tpw_out) *tpw_out = 1;
    if (max_n <= 0) return 0;
    const int route = vh_lk_route(batch, max_n, win);
    if (route_out) *route_out = route;
    switch (route) {
    case 3: return launch_lk3<15, 1, 6>(job_tab, tab_stride, batch, max_n, s, tpw_out);
    case 5: return launch_lk3<51, 1, 4>(job_tab, tab_stride, batch, max_n, s, tpw_out);
    case 6: return launch_lk3<51, 2, 4>(job_tab, tab_stride, batch, max_n, s, tpw_out);
    case 7: return launch_lk3<51, 4, 4>(job_tab, tab_stride, batch, max_n, s, tpw_out);
    case 8: return launch_lko<15>(job_tab, tab_stride, batch, max_n, s);
    case 4: return launch_lkq<15>(job_tab, tab_stride, batch, max_n, s);
    case 2:
        if (win == 15) return launch_strip<15>(job_tab, tab_stride, batch, max_n, win, s);
        if (win == 51) return launch_strip<51>(job_tab, tab_stride, batch, max_n, win, s);
        return launch_strip<0>(job_tab, tab_stride, batch, max_n, win, s);
    default: break;
    }
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
