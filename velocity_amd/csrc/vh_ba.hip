// Bundle adjustment (K14): fcnNLS_batch (utils/NLS.py:186-250) -- dense Levenberg-Marquardt over all tie points and
// cameras 1..nc with the constant +I damping, step 0.9, <= 10 iterations.
//
// The reference materialises the dense forward-difference J^T (nx x nz) and inverts J^T J + I.  J has 9 non-zeros per
// row (3 for the point, 6 for the camera), so here the forward-difference entries are computed exactly as the reference
// defines them (same dx = 1e-6 perturbations of x, same re-projection) but kept compact, and the damped normal equations
// are solved through the point-block Schur complement -- algebraically the same delta as the dense inverse:
//     H = [[U  W],[W^T V]] + I,   S = V + I - W^T (U+I)^-1 W,   S dc = gc - W^T (U+I)^-1 gp,   dp = (U+I)^-1 (gp - W dc)
// Everything stays on the device for all iterations (the convergence test only sets a device flag).
#include <atomic>
#include "vh_ba.hpp"
#include <type_traits>
#include "vh_ws.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

#define BA_FD 1e-6
#define BA_THREADS 256
#define BA_EPT 64  // Schur entries owned by one thread per pass
#define BA_MAX_NC 255                       // free cameras: a k_ba_jac workgroup (256 threads) owns whole points, one thread per frame (the reference has no limit)
#define BA_MAX_NC_VALU 128                  // ... of the VALU Schur kernel (test hook / fcnNLS_batch2's trajectory model excepted: nc + 5 unknowns)
#define BA_ZB_MAX 256                        // workgroups of k_ba_zbuild (43+ cameras): one partial right-hand side / set of diagonal blocks each
#define BA_RQ ((6 * BA_MAX_NC_VALU + BA_THREADS - 1) / BA_THREADS)  // reduced right-hand-side entries owned by one thread of the VALU Schur kernel

// batched windows: shift every pointer of the job to window w (the job travels by value, so this edits the kernel's own copy)
__device__ __forceinline__ void ba_select_window(BaJob& J, int w)
{
    if (w == 0) return;
    const size_t d = (size_t)w * (J.ws_stride / sizeof(double));
    J.camR += d; J.r += d; J.Jp += d; J.Jc += d; J.tp += d; J.Lc += d; J.Y += d; J.Spart += d; J.Rpart += d; J.Dpart += d; J.Sfull += d; J.dc += d; J.acc += d; J.rslot += d;
    J.done = reinterpret_cast<int*>(reinterpret_cast<char*>(J.done) + (size_t)w * J.ws_stride);
    J.ticket = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(J.ticket) + (size_t)w * J.ws_stride);
    J.z += (size_t)w * J.z_stride; J.x += (size_t)w * J.x_stride; J.trace += (size_t)w * J.trace_stride; J.info += (size_t)w * J.info_stride;
}

__device__ void ba_rpy2dcm(const double* rpy, double* C)
{
    const double sr = sin(rpy[0]), cr = cos(rpy[0]), sp = sin(rpy[1]), cp = cos(rpy[1]), sy = sin(rpy[2]), cy = cos(rpy[2]);
    C[0] = cp * cy; C[1] = sr * sp * cy - cr * sy; C[2] = cr * sp * cy + sr * sy;
    C[3] = cp * sy; C[4] = sr * sp * sy + cr * cy; C[5] = cr * sp * sy - sr * cy;
    C[6] = -sp;     C[7] = sr * cp;                C[8] = cr * cp;
}

__device__ __forceinline__ void ba_project(const double* K, const double* R, const double* w, const double* t, double& u, double& v)
{
    const double b0 = w[0] * R[0] + w[1] * R[3] + w[2] * R[6] + t[0];
    const double b1 = w[0] * R[1] + w[1] * R[4] + w[2] * R[7] + t[1];
    const double b2 = w[0] * R[2] + w[1] * R[5] + w[2] * R[8] + t[2];
    const double q0 = b0 * K[0] + b1 * K[3] + b2 * K[6];
    const double q1 = b0 * K[1] + b1 * K[4] + b2 * K[7];
    const double q2 = b0 * K[2] + b1 * K[5] + b2 * K[8];
    // u = q0 / q2, v = q1 / q2 (pscale, NLS.py:78) with ONE division: r = RN(1 / q2), then q = a r, e = fma(-q, q2, a), a / q2 = fma(e, r, q) -- correctly
    // rounded like the division itself (Markstein; 0 mismatches in 2e5 random cases against exact rational arithmetic), 17 instructions instead of 22
    const double r = 1.0 / q2;
    const double uq = q0 * r, vq = q1 * r;
    u = __builtin_fma(__builtin_fma(-uq, q2, q0), r, uq);
    v = __builtin_fma(__builtin_fma(-vq, q2, q1), r, vq);
    // q2 == 0 (a point on the camera plane) or an overflowing / vanishing reciprocal: the correction would give fma(-inf, 0, q0) = NaN where the
    // division -- and numpy's pscale -- give +-inf / 0, and the NaN would spread through the whole window's Schur system.  Rare: a branch, not a select.
    if (__builtin_expect(!(r != 0.0 && __builtin_isfinite(r)), 0)) { u = q0 / q2; v = q1 / q2; }
}

// x / BA_FD (the reference's forward difference (f(x + dx) - f(x)) / dx, NLS.py:228-233) without the division: RN(1 / 1e-6) is exactly 1e6, and
// q = x 1e6, e = fma(-q, 1e-6, x), fma(e, 1e6, q) is the correctly rounded quotient (0 mismatches in 2e5 random cases over 50 binades against exact
// rational arithmetic): 3 instructions instead of the 11 of an IEEE division, 18 of them per measurement
__device__ __forceinline__ double ba_div_fd(double x)
{
    const double q = x * 1.0e6;
    return __builtin_fma(__builtin_fma(-q, BA_FD, x), 1.0e6, q);
}

// camera rotation matrices: R(rpy) and the three forward-difference neighbours R(rpy + dx e_k)  (NLS.py:206-216,228-233)
// model 1 (fcnNLS_batch2, NLS.py:278-291): one joint rotation applied to the points; camera c sits at
// sc2cc([range_c, el, az]) @ cam2ned() on a straight line; camera 0 is the origin
__device__ void ba2_offset(double range, double el, double az, double* o)
{
    // sc2cc (common.py:97-112): ned = [r cos(el) cos(az), r cos(el) sin(az), -r sin(el)];  ned @ [[0,0,1],[1,0,0],[0,1,0]] = [n1, n2, n0]
    const double a = range * cos(el);
    const double n0 = a * cos(az), n1 = a * sin(az), n2 = -range * sin(el);
    o[0] = n1; o[1] = n2; o[2] = n0;
}

// par = the camera-side part of the state (J.x + 3 nt, or a copy of it): model 0 [positions (3 nc) | rpy (3 nc)], model 1 [rpy, el, az, ranges]
__device__ void ba_cam_tables(const BaJob& J, int c, const double* par)  // camera index 0..nc (0 = fixed identity camera)
{
    if (J.model == 1) {
        const double* g = par;  // rpy(3), el, az, ranges(nc)
        if (c == 0) {
            ba_rpy2dcm(g, J.camR);
            for (int k = 0; k < 3; k++) {
                double a[3] = {g[0], g[1], g[2]};
                a[k] += BA_FD;
                ba_rpy2dcm(a, J.camR + 9 * (k + 1));
            }
            for (int k = 0; k < 12; k++) J.camR[36 + k] = 0.0;
            return;
        }
        double* o = J.camR + 36 + 12 * (size_t)c;
        const double rg = g[5 + (c - 1)], el = g[3], az = g[4];
        ba2_offset(rg, el, az, o);
        ba2_offset(rg, el + BA_FD, az, o + 3);
        ba2_offset(rg, el, az + BA_FD, o + 6);
        ba2_offset(rg + BA_FD, el, az, o + 9);
        return;
    }
    double* out = J.camR + (size_t)c * 36;
    if (c == 0) {
        for (int q = 0; q < 4; q++)
            for (int k = 0; k < 9; k++) out[q * 9 + k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double* rpy = par + 3 * J.nc + 3 * (c - 1);
    ba_rpy2dcm(rpy, out);
    for (int k = 0; k < 3; k++) {
        double a[3] = {rpy[0], rpy[1], rpy[2]};
        a[k] += BA_FD;
        ba_rpy2dcm(a, out + 9 * (k + 1));
    }
}

// model 0, one (camera, variant) pair: variant 0 = R(rpy), 1..3 = R(rpy + dx e_k) -- the four rotation matrices of a camera are independent chains of six
// f64 sin / cos each; one thread per pair keeps the chain of the update kernel's camera block four times shorter
__device__ void ba_cam_table_variant(const BaJob& J, int c, int v, const double* par)
{
    double* out = J.camR + (size_t)c * 36 + 9 * v;
    if (c == 0) {
        for (int k = 0; k < 9; k++) out[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double* rpy = par + 3 * J.nc + 3 * (c - 1);
    double a[3] = {rpy[0], rpy[1], rpy[2]};
    if (v > 0) a[v - 1] += BA_FD;
    ba_rpy2dcm(a, out);
}

// first iteration only: later iterations get their tables from block 0 of k_ba_update, right after it moved the cameras
__global__ void k_ba_cams(BaJob J)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= J.nc) ba_cam_tables(J, c, J.x + 3 * J.nt);  // (once per solve: one thread per camera is fine here)
}

__device__ void inv3_sym(const double* U, double* Ui);

// residual and compact forward-difference Jacobian of every measurement pair (camera c, track i).
// A thread owns one measurement (20 doubles of output: r 2, Jp 6, Jc 12).  Written straight from the registers, a wave's store touched 64
// separate 16 / 48 / 96-byte records per instruction; the values are transposed through LDS instead and leave as fully coalesced streams
// (the block's measurements are contiguous in all three arrays).  A block owns WHOLE points (256 / (nc+1) of them, the last threads idle), so
// with PREP it also finishes what the point-block Schur complement needs per point -- U_i + I, its inverse, tp_i, the Cholesky factor of the
// inverse -- straight from the LDS copy of the rows: the separate pass re-read 8 of the 20 planes from HBM (0.8 GB per iteration at 64 C5 windows).
template <bool PREP>
__global__ __launch_bounds__(BA_THREADS) void k_ba_jac(BaJob J)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nt = J.nt, nf = J.nc + 1;
    const int ppb = BA_THREADS / nf, i0 = blockIdx.x * ppb, npts = min(ppb, nt - i0);
    const int il = threadIdx.x / nf;
    __shared__ double s_out[20][BA_THREADS + 1];  // component-major, padded: conflict-free on the way in, spread on the way out
    double ss = 0.0;
    double o[20];  // r (2) | Jp (6) | Jc (12)
#pragma unroll
    for (int k = 0; k < 20; k++) o[k] = 0.0;
    if (il < npts) {
        const int i = i0 + il, c = threadIdx.x - il * nf;  // POINT-major measurement index m = i (nc+1) + c: a point's Jacobians are contiguous
        double K[9];
        for (int k = 0; k < 9; k++) K[k] = J.K[k];
        double w[3] = {J.x[3 * i], J.x[3 * i + 1], J.x[3 * i + 2]};
        if (J.model == 1) {
            // zhat = pscale((pw @ R + offset_c) @ K); columns of the compact Jacobian: point (3) | rpy (3), el, az, range_c
            const double* R0 = J.camR;
            const double* off = J.camR + 36 + 12 * (size_t)c;
            double u, v, uk, vk;
            ba_project(K, R0, w, off, u, v);
            const size_t mz = (size_t)c * nt + i;  // z stays in the reference's order: [all u | all v], camera-major (NLS.py:198-199)
            const double ru = J.z[mz] - u, rv = J.z[(size_t)nt * nf + mz] - v;
            o[0] = ru; o[1] = rv;
            ss = ru * ru + rv * rv;
            for (int k = 0; k < 3; k++) {
                double wk[3] = {w[0], w[1], w[2]};
                wk[k] += BA_FD;
                ba_project(K, R0, wk, off, uk, vk);
                o[2 + k] = ba_div_fd(uk - u);
                o[5 + k] = ba_div_fd(vk - v);
            }
            for (int k = 0; k < 3; k++) {  // joint roll / pitch / yaw
                ba_project(K, R0 + 9 * (k + 1), w, off, uk, vk);
                o[8 + k] = ba_div_fd(uk - u);
                o[14 + k] = ba_div_fd(vk - v);
            }
            for (int k = 0; k < 3; k++) {  // el, az, range of this camera (camera 0 is fixed: exact zeros, as the reference's FD gives)
                if (c > 0) {
                    ba_project(K, R0, w, off + 3 * (k + 1), uk, vk);
                    o[11 + k] = ba_div_fd(uk - u);
                    o[17 + k] = ba_div_fd(vk - v);
                }
            }
        } else {
            const double* R = J.camR + (size_t)c * 36;
            double t[3] = {0, 0, 0};
            if (c > 0) for (int k = 0; k < 3; k++) t[k] = J.x[3 * nt + 3 * (c - 1) + k];
            double u, v, uk, vk;
            ba_project(K, R, w, t, u, v);
            const size_t mz = (size_t)c * nt + i;  // z = [all u | all v], camera-major (NLS.py:198-199)
            const double ru = J.z[mz] - u, rv = J.z[(size_t)nt * nf + mz] - v;
            o[0] = ru; o[1] = rv;
            ss = ru * ru + rv * rv;
            for (int k = 0; k < 3; k++) {  // point coordinates
                double wk[3] = {w[0], w[1], w[2]};
                wk[k] += BA_FD;
                ba_project(K, R, wk, t, uk, vk);
                o[2 + k] = ba_div_fd(uk - u);
                o[5 + k] = ba_div_fd(vk - v);
            }
            if (c > 0) {
                for (int k = 0; k < 3; k++) {  // camera position
                    double tk[3] = {t[0], t[1], t[2]};
                    tk[k] += BA_FD;
                    ba_project(K, R, w, tk, uk, vk);
                    o[8 + k] = ba_div_fd(uk - u);
                    o[14 + k] = ba_div_fd(vk - v);
                }
                for (int k = 0; k < 3; k++) {  // camera roll / pitch / yaw
                    ba_project(K, R + 9 * (k + 1), w, t, uk, vk);
                    o[11 + k] = ba_div_fd(uk - u);
                    o[17 + k] = ba_div_fd(vk - v);
                }
            }  // camera 0 is fixed: its 12 entries stay exact zeros
        }
    }
    // a NaN measurement (no observation of this track in this frame) takes no part: zero residual, zero Jacobian rows.  (The reference zeroes
    // z and zhat there, NLS.py:200-201,225, but leaves f(x + dx e_j) - 0 in its forward differences: rows of ~1e9.  Resolved by intent; with the
    // full-length track filter of NLS.py:190 it only happens when P[0:2] is NaN where P[4] is not.)
    if (ss != ss) {
        ss = 0.0;
#pragma unroll
        for (int k = 0; k < 20; k++) o[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 20; k++) s_out[k][threadIdx.x] = o[k];
    // sum of squared residuals of this iteration (trace only): one atomic per block, spread over 16 addresses -- thousands of
    // same-address atomics would serialise in L2 and dominate the kernel
    __shared__ double s_ss[BA_THREADS / 64];
    ss = vh_wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) s_ss[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < BA_THREADS / 64; k++) t += s_ss[k];
        if (t != 0.0) atomicAdd(J.rslot + (blockIdx.x & 15), t);
    }
    // coalesced write-out of the block's contiguous spans: r [2 x nval], Jp [6 x nval], Jc [12 x nval] doubles
    // (round 5 ablation, one C5 window: 17.1 us in all = 6.1 launch / LDS / atomics + 1 arithmetic + 8.6 the 16 MB of stores and their write-back + 1.4 the tail below)
    const size_t m0 = (size_t)i0 * nf;
    const int nval = npts * nf;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int e = tid + BA_THREADS * q, t = e >> 1, k = e & 1;
        if (t < nval) J.r[2 * m0 + e] = s_out[k][t];
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const int e = tid + BA_THREADS * q, t = e / 6, k = e - 6 * t;
        if (t < nval) J.Jp[6 * m0 + e] = s_out[2 + k][t];
    }
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const int e = tid + BA_THREADS * q, t = e / 12, k = e - 12 * t;
        if (t < nval) J.Jc[12 * m0 + e] = s_out[8 + k][t];
    }
    if (!PREP) return;
    // per point (4 lanes each, measurements c = q, q+4, ...): U_i = I + sum Jp^T Jp, g_i = sum Jp^T r  ->  tp_i = U^-1 g, U^-1 = L L^T
    // a block owns ppb = 256 / (nc+1) points but only 64 quads: 2- and 3-frame windows (ppb = 128, 85) take two rounds
    const int q = tid & 3;
    for (int pl = tid >> 2; pl < npts; pl += BA_THREADS / 4) {  // whole quads stay together
    double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0, u4 = 0.0, u5 = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
    for (int c = q; c < nf; c += 4) {
        const int t = pl * nf + c;
        const double ru = s_out[0][t], rv = s_out[1][t], a0 = s_out[2][t], a1 = s_out[3][t], a2 = s_out[4][t], b0 = s_out[5][t], b1 = s_out[6][t], b2 = s_out[7][t];
        u0 += a0 * a0 + b0 * b0; u1 += a0 * a1 + b0 * b1; u2 += a0 * a2 + b0 * b2;
        u3 += a1 * a1 + b1 * b1; u4 += a1 * a2 + b1 * b2; u5 += a2 * a2 + b2 * b2;
        g0 += a0 * ru + b0 * rv; g1 += a1 * ru + b1 * rv; g2 += a2 * ru + b2 * rv;
    }
    double acc[9] = {u0, u1, u2, u3, u4, u5, g0, g1, g2};
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc[k] += __shfl_xor(acc[k], 1);
        acc[k] += __shfl_xor(acc[k], 2);
    }
    if (q != 0) continue;
    const int i = i0 + pl;
    const double U[9] = {acc[0] + 1.0, acc[1], acc[2], acc[1], acc[3] + 1.0, acc[4], acc[2], acc[4], acc[5] + 1.0};  // +I damping (NLS.py:220)
    double Ui[9];
    inv3_sym(U, Ui);
    J.tp[3 * (size_t)i] = Ui[0] * acc[6] + Ui[1] * acc[7] + Ui[2] * acc[8];
    J.tp[3 * (size_t)i + 1] = Ui[3] * acc[6] + Ui[4] * acc[7] + Ui[5] * acc[8];
    J.tp[3 * (size_t)i + 2] = Ui[6] * acc[6] + Ui[7] * acc[7] + Ui[8] * acc[8];
    // Cholesky of the SPD inverse: Ui = L L^T
    const double l00 = sqrt(Ui[0]), l10 = Ui[3] / l00, l20 = Ui[6] / l00;
    const double l11 = sqrt(Ui[4] - l10 * l10), l21 = (Ui[7] - l20 * l10) / l11;
    const double l22 = sqrt(Ui[8] - l20 * l20 - l21 * l21);
    double* Lo = J.Lc + 6 * (size_t)i;
    Lo[0] = l00; Lo[1] = l10; Lo[2] = l11; Lo[3] = l20; Lo[4] = l21; Lo[5] = l22;
    }
}

__device__ void inv3_sym(const double* U, double* Ui)
{
    const double a = U[0], b = U[1], c = U[2], d = U[4], e = U[5], f = U[8];
    const double A = d * f - e * e, B = -(b * f - c * e), C = b * e - c * d;
    const double det = a * A + b * B + c * C;
    const double id = 1.0 / det;
    Ui[0] = A * id; Ui[1] = B * id; Ui[2] = C * id;
    Ui[3] = B * id; Ui[4] = (a * f - c * c) * id; Ui[5] = -(a * e - b * c) * id;
    Ui[6] = C * id; Ui[7] = Ui[5]; Ui[8] = (a * d - b * b) * id;
}

// Schur stage 1: one workgroup per chunk of points.  Per point: U_i, its inverse, W_i, Y_i = U_i^-1 W_i, tp_i = U_i^-1 gp_i;
// the workgroup's partial of S (thread-owned entries, registers) and of the reduced right-hand side go to global memory.
__global__ __launch_bounds__(BA_THREADS) void k_ba_points(BaJob J, int pass)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = J.nq, tid = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sW = reinterpret_cast<double*>(smem);  // [3][nq]
    double* sY = sW + 3 * nq;                      // [3][nq]
    double* sJc = sY + 3 * nq;                     // [nc + 1][12]
    double* sTp = sJc + 12 * (nc + 1);             // [3] tp_i, [3..] scratch
    const int chunk = (nt + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(nt, i0 + chunk);
    const long long nent = (long long)nq * nq;
    const long long ebase = (long long)pass * BA_THREADS * BA_EPT;
    double accS[BA_EPT];
#pragma unroll
    for (int e = 0; e < BA_EPT; e++) accS[e] = 0.0;
    double accR[BA_RQ];  // reduced rhs entries tid + 256 j (first pass only)
#pragma unroll
    for (int j = 0; j < BA_RQ; j++) accR[j] = 0.0;

    for (int i = i0; i < i1; i++) {
        __syncthreads();
        // thread 0..: U_i and gp_i over all cameras (tiny) -- done redundantly by the first wave's lane 0
        if (tid == 0) {
            double U[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, gp[3] = {0, 0, 0};  // +I damping (NLS.py:220)
            for (int c = 0; c <= nc; c++) {
                const size_t m = (size_t)i * (nc + 1) + c;
                const double* Jp = J.Jp + 6 * m;
                const double ru = J.r[2 * m], rv = J.r[2 * m + 1];
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++) U[a * 3 + b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
                    gp[a] += Jp[a] * ru + Jp[3 + a] * rv;
                }
            }
            double Ui[9];
            inv3_sym(U, Ui);
            for (int k = 0; k < 9; k++) sTp[3 + k] = Ui[k];
            for (int a = 0; a < 3; a++) sTp[a] = Ui[a * 3] * gp[0] + Ui[a * 3 + 1] * gp[1] + Ui[a * 3 + 2] * gp[2];
            if (pass == 0) {
                for (int a = 0; a < 3; a++) J.tp[3 * (size_t)i + a] = sTp[a];
            }
        }
        // stage the camera Jacobians of this point (model 0: cameras 1..nc at sJc[12 (c-1)]; model 1: cameras 0..nc at sJc[12 c])
        if (J.model == 1) {
            for (int q = tid; q < 12 * (nc + 1); q += BA_THREADS) {
                const int c = q / 12, k = q - c * 12;
                sJc[q] = J.Jc[12 * ((size_t)i * (nc + 1) + c) + k];
            }
        } else {
            for (int q = tid; q < 12 * nc; q += BA_THREADS) {
                const int c = q / 12 + 1, k = q - (c - 1) * 12;
                sJc[q] = J.Jc[12 * ((size_t)i * (nc + 1) + c) + k];
            }
        }
        __syncthreads();
        // W_i [3][nq] and Y_i = U_i^-1 W_i
#pragma unroll
        for (int jq = 0; jq < BA_RQ; jq++) {
            const int q = tid + BA_THREADS * jq;
            if (q >= nq) break;
            double w0 = 0.0, w1 = 0.0, w2 = 0.0, gq = 0.0;
            if (J.model == 1) {
                // column q < 5 (joint rpy, el, az) collects every camera, column q >= 5 is the range of camera q - 4
                const int c_lo = q < 5 ? 0 : q - 4, c_hi = q < 5 ? nc : q - 4, k = q < 5 ? q : 5;
                for (int c = c_lo; c <= c_hi; c++) {
                    const size_t m = (size_t)i * (nc + 1) + c;
                    const double* Jp = J.Jp + 6 * m;
                    const double ju = sJc[12 * c + k], jv = sJc[12 * c + 6 + k];
                    w0 += Jp[0] * ju + Jp[3] * jv; w1 += Jp[1] * ju + Jp[4] * jv; w2 += Jp[2] * ju + Jp[5] * jv;
                    gq += ju * J.r[2 * m] + jv * J.r[2 * m + 1];
                }
            } else {
                const int c = q / 6, k = q - 6 * c;  // camera c+1, parameter k (0..2 pos, 3..5 rpy)
                const size_t m = (size_t)i * (nc + 1) + (c + 1);
                const double* Jp = J.Jp + 6 * m;
                const double ju = sJc[12 * c + k], jv = sJc[12 * c + 6 + k];
                w0 = Jp[0] * ju + Jp[3] * jv; w1 = Jp[1] * ju + Jp[4] * jv; w2 = Jp[2] * ju + Jp[5] * jv;
                gq = ju * J.r[2 * m] + jv * J.r[2 * m + 1];
            }
            sW[q] = w0; sW[nq + q] = w1; sW[2 * nq + q] = w2;
            const double* Ui = sTp + 3;
            const double y0 = Ui[0] * w0 + Ui[1] * w1 + Ui[2] * w2, y1 = Ui[3] * w0 + Ui[4] * w1 + Ui[5] * w2, y2 = Ui[6] * w0 + Ui[7] * w1 + Ui[8] * w2;
            sY[q] = y0; sY[nq + q] = y1; sY[2 * nq + q] = y2;
            if (pass == 0) {
                double* Yg = J.Y + ((size_t)i * nq + q) * 3;
                Yg[0] = y0; Yg[1] = y1; Yg[2] = y2;
                // reduced rhs: gc - W^T tp
                accR[jq] += gq - (w0 * sTp[0] + w1 * sTp[1] + w2 * sTp[2]);
            }
        }
        __syncthreads();
        // thread-owned entries of S: [c == c'] Jc^T Jc - W^T Y
#pragma unroll
        for (int e = 0; e < BA_EPT; e++) {
            const long long ent = ebase + (long long)e * BA_THREADS + tid;
            if (ent < nent) {
                const int a = (int)(ent / nq), b = (int)(ent - (long long)a * nq);
                double v = -(sW[a] * sY[b] + sW[nq + a] * sY[nq + b] + sW[2 * nq + a] * sY[2 * nq + b]);
                if (J.model == 1) {
                    // V[a][b] over the cameras that carry both columns (shared columns: all; a range column: its camera)
                    const int ka = a < 5 ? a : 5, kb = b < 5 ? b : 5;
                    int c_lo = 0, c_hi = nc;
                    if (a >= 5) { c_lo = a - 4; c_hi = a - 4; }
                    if (b >= 5) { c_lo = max(c_lo, b - 4); c_hi = min(c_hi, b - 4); }
                    for (int c = c_lo; c <= c_hi; c++) v += sJc[12 * c + ka] * sJc[12 * c + kb] + sJc[12 * c + 6 + ka] * sJc[12 * c + 6 + kb];
                    accS[e] += v;
                    continue;
                }
                const int ca = a / 6, cb = b / 6;
                if (ca == cb) {
                    const int ka = a - 6 * ca, kb = b - 6 * cb;
                    v += sJc[12 * ca + ka] * sJc[12 * ca + kb] + sJc[12 * ca + 6 + ka] * sJc[12 * ca + 6 + kb];
                }
                accS[e] += v;
            }
        }
    }
    double* Sp = J.Spart + (size_t)blockIdx.x * nent;
#pragma unroll
    for (int e = 0; e < BA_EPT; e++) {
        const long long ent = ebase + (long long)e * BA_THREADS + tid;
        if (ent < nent) Sp[ent] = accS[e];
    }
    if (pass == 0) {
#pragma unroll
        for (int jq = 0; jq < BA_RQ; jq++)
            if (tid + BA_THREADS * jq < nq) J.Rpart[(size_t)blockIdx.x * nq + tid + BA_THREADS * jq] = accR[jq];
    }
}

// ---- Schur stage 1 on the matrix cores (6 nc <= 128) ------------------------------------------------------------------------------
// (a) the tail of k_ba_jac<true>, four lanes per tie point: U_i + I over all cameras, its inverse, tp_i = (U_i+I)^-1 gp_i and the Cholesky factor L_i of the
//     inverse ((U_i+I)^-1 = L L^T).  Tiny, but it takes the per-point reduction, the 3x3 inversion and their dependent-latency chain out of
//     the matrix-core kernel.
// (b) k_ba_schur_mfma: with Z_i = L_i^T W_i the reduced camera system is S = V + I - sum_i Z_i^T Z_i -- a SYMMETRIC rank-k update (the genuine
//     dense contraction of BA, (6nc x 3nt) . (3nt x 6nc)): only the 36 upper-triangle tiles of the 8 x 8 grid of 16 x 16 tiles are computed,
//     A and B operands of v_mfma_f64_16x16x4_f64 are the same LDS rows.  One workgroup (4 wavefronts) per chunk of points, points in groups of
//     4 (12 rows of Z, double-buffered in LDS: ONE barrier per group); wavefront w owns tile rows w and 7-w (9 tiles, 72 accumulator
//     registers), so two workgroups fit a CU and one's global-load latency hides behind the other's matrix-core time.
//     Z is also what the back-substitution needs: dp_i = tp_i - L_i (Z_i dc).
typedef double double4v __attribute__((ext_vector_type(4)));
#define BA_NPAD 128

// ---- k_ba_schur_mfma: data flow --------------------------------------------------------------------------------------------------------
// The inputs of one tie point ("raw record": its camera Jacobians Jc [nc][12], point Jacobians Jp [nc][6], residuals r [nc][2], L (6), tp (3);
// 20 nc + 9 doubles) are fetched ONCE per lane-word (7 global loads per lane and point, no duplicate addresses), two groups ahead of their use,
// and parked in LDS (double buffered).  A group = 4 points (one per wavefront).  Per group and wavefront, in ONE basic block:
//     matrix cores : the 9 upper-triangle tiles of Z_g^T Z_g (3 K-slabs, 27 v_mfma_f64_16x16x4_f64) from the LDS copy of Z_g
//     VALU / LDS   : Z_{g+1} = L^T W of the NEXT group from its raw record (-> the other Z buffer), the reduced right-hand side and the
//                    thread-owned entries of the diagonal blocks V_c of the next group
// so the vector work of a group hides behind the matrix-core time of the previous one, and there is exactly one workgroup barrier per group.
// Two sizes of the same kernel (template NP = padded width of the reduced system): NP = 128 (up to 21 cameras: an 8 x 8 tile grid, 36 upper-triangle tiles,
// ONE launch) and, round 4, NP = 256 (22..42 cameras: 16 x 16 tiles, 136 upper-triangle tiles in TWO launches of 68 -- a consumer wavefront then holds 17
// tiles = 136 accumulator registers; both launches rebuild Z, which is the cheap part).  Tile rows are paired (W, NT-1-W): NT+1 tiles per wavefront job.
template <int NP>
struct BaSchurCfg {
    static constexpr int NT = NP / 16;               // tile rows / columns
    static constexpr int NH = NP / 64;               // reduced columns per producer lane
    static constexpr int NRW = NP == 128 ? 7 : 14;   // lane-words of a raw record: ceil((20 nc + 9) / 64) for nc <= 21 / 42
    static constexpr int ND = NP == 128 ? 3 : 6;     // diagonal-block entries per producer thread: ceil(36 nc / 256)
    static constexpr int NACC = NT + 1;              // tiles of one consumer job
};

template <int NRW>
struct BaRawOff {                 // loop-invariant per lane: where its words of a raw record live in global memory
    const double* base[NRW];
    int stride[NRW];              // doubles per point (0: lane has no such word)
};

template <int NRW>
__device__ __forceinline__ void ba_raw_offsets(BaRawOff<NRW>& O, const BaJob& J, int lane)
{
    const int nc = J.nc, nw = 20 * nc + 9;
#pragma unroll
    for (int j = 0; j < NRW; j++) {
        const int w = min(lane + 64 * j, nw - 1);  // clamped: surplus lanes re-load the last word (never stored)
        const double* b;
        int st;
        // point-major arrays: the words of cameras 1..nc of one point are CONTIGUOUS (camera 0 comes first and is skipped): fully coalesced loads
        if (w < 12 * nc) { b = J.Jc + 12 + w; st = 12 * (nc + 1); }
        else if (w < 18 * nc) { b = J.Jp + 6 + (w - 12 * nc); st = 6 * (nc + 1); }
        else if (w < 20 * nc) { b = J.r + 2 + (w - 18 * nc); st = 2 * (nc + 1); }
        else if (w < 20 * nc + 6) { b = J.Lc + (w - 20 * nc); st = 6; }
        else { b = J.tp + (w - 20 * nc - 6); st = 3; }
        O.base[j] = b;
        O.stride[j] = st;
    }
}

// the lane's words of the raw record of point min(i, i_last) (branch-free: unconditional loads, all in flight together)
template <int NRW>
__device__ __forceinline__ void ba_raw_fetch(double (&R)[NRW], const BaRawOff<NRW>& O, int i, int i_last)
{
    i = min(i, i_last);
#pragma unroll
    for (int j = 0; j < NRW; j++) R[j] = O.base[j][(size_t)O.stride[j] * i];
}

template <int NRW>
__device__ __forceinline__ void ba_raw_park(const double (&R)[NRW], double* __restrict__ rec, int lane, int nwords, bool live)
{
#pragma unroll
    for (int j = 0; j < NRW; j++) {
        const int w = lane + 64 * j;
        if (w < nwords) rec[w] = live ? R[j] : 0.0;  // a dead point (past the chunk) is all zeros: it adds nothing anywhere
    }
}

// per-lane LDS offsets (doubles, inside a raw record) of what the Z rows of column q = lane + 64 h need
template <int NH, int ND>
struct BaColOff {
    int jp[NH], ju[NH], rr[NH];
    int da[ND], db[ND];  // diagonal-block entry e = tid + 256 k of this thread: offsets 12 c + ka, 12 c + kb inside a record (clamped)
};

// consumer job W (matrix cores): the 3 K-slabs of its NT + 1 upper-triangle tiles (tile row W: columns W..NT-1, tile row NT-1-W: columns NT-1-W..NT-1)
template <int NP, int W>
__device__ __forceinline__ void ba_consume(double4v (&acc)[NP / 16 + 1], const double* __restrict__ sZc, int lane)
{
    constexpr int NT = NP / 16, R1 = W, R2 = NT - 1 - W, T0 = R1 < R2 ? R1 : R2;
    const int cc = lane & 15;
#pragma unroll
    for (int k0 = 0; k0 < 12; k0 += 4) {
        const int kr = k0 + (lane >> 4);
        if constexpr (NT <= 8) {
            double zf[NT];
#pragma unroll
            for (int t = T0; t < NT; t++) zf[t] = sZc[kr * NP + 16 * t + cc];
#pragma unroll
            for (int t = R1; t < NT; t++) acc[t - R1] = __builtin_amdgcn_mfma_f64_16x16x4f64(zf[R1], zf[t], acc[t - R1], 0, 0, 0);
#pragma unroll
            for (int t = R2; t < NT; t++) acc[NT - R1 + t - R2] = __builtin_amdgcn_mfma_f64_16x16x4f64(zf[R2], zf[t], acc[NT - R1 + t - R2], 0, 0, 0);
        } else {
            // 16 tile columns: the operand of a tile is read right before its instruction (all 16 at once would be 32 more live registers next to the
            // 136 accumulator registers); the tiles of row R2 read their columns a second time
            const double a1 = sZc[kr * NP + 16 * R1 + cc], a2 = sZc[kr * NP + 16 * R2 + cc];
#pragma unroll
            for (int t = R1; t < NT; t++) {
                const double b = t == R1 ? a1 : (t == R2 ? a2 : sZc[kr * NP + 16 * t + cc]);
                acc[t - R1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[t - R1], 0, 0, 0);
            }
#pragma unroll
            for (int t = R2; t < NT; t++) {
                const double b = t == R2 ? a2 : sZc[kr * NP + 16 * t + cc];
                acc[NT - R1 + t - R2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc[NT - R1 + t - R2], 0, 0, 0);
            }
        }
    }
}

// producer wavefront pw: Z rows 3 pw .. 3 pw + 2 of its point from the point's raw record (already parked in LDS by this very wavefront),
// plus the point's share of the reduced right-hand side
template <int NP>
__device__ __forceinline__ void ba_produce_z(double (&accR)[NP / 64], double* __restrict__ sZn, const double* __restrict__ rec,
                                             const BaColOff<NP / 64, BaSchurCfg<NP>::ND>& C, int nc, int nq, int lane, int pw)
{
    const double* lt = rec + 20 * nc;  // L, tp: wave-uniform
    const double L0 = lt[0], L1 = lt[1], L2 = lt[2], L3 = lt[3], L4 = lt[4], L5 = lt[5], t0 = lt[6], t1 = lt[7], t2 = lt[8];
#pragma unroll
    for (int h = 0; h < NP / 64; h++) {
        const double* jp = rec + C.jp[h];
        const double ju = rec[C.ju[h]], jv = rec[C.ju[h] + 6], ru = rec[C.rr[h]], rv = rec[C.rr[h] + 1];
        const double w0 = jp[0] * ju + jp[3] * jv, w1 = jp[1] * ju + jp[4] * jv, w2 = jp[2] * ju + jp[5] * jv;
        // L^T is upper triangular: rows (l00 l10 l20), (0 l11 l21), (0 0 l22); L is stored l00 l10 l11 l20 l21 l22
        const double z0 = L0 * w0 + L1 * w1 + L3 * w2, z1 = L2 * w1 + L4 * w2, z2 = L5 * w2;
        accR[h] += ju * ru + jv * rv - (w0 * t0 + w1 * t1 + w2 * t2);  // a dead point's record is zeros; lanes with q >= nq hold a duplicate that is never stored
        const int q = lane + 64 * h;  // < NP = the row pitch: columns >= nq are written as zeros
        const bool on = q < nq;
        sZn[(3 * pw) * NP + q] = on ? z0 : 0.0; sZn[(3 * pw + 1) * NP + q] = on ? z1 : 0.0; sZn[(3 * pw + 2) * NP + q] = on ? z2 : 0.0;
    }
}

// producer threads: their entries of the diagonal blocks V_c, summed over the 4 points of a group (records of all four producers)
template <int NH, int ND>
__device__ __forceinline__ void ba_produce_diag(double (&accD)[ND], const double* __restrict__ raw4, int RAWW, const BaColOff<NH, ND>& C)
{
#pragma unroll
    for (int k = 0; k < ND; k++) {
        double v = 0.0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const double* ja = raw4 + g * RAWW + C.da[k];
            const double* jb = raw4 + g * RAWW + C.db[k];
            v += ja[0] * jb[0] + ja[6] * jb[6];
        }
        accD[k] += v;
    }
}

template <int NP, int W>
__device__ __forceinline__ void ba_syrk_store(const double4v (&acc)[NP / 16 + 1], double* __restrict__ Sp, int lane, int nq)
{
    constexpr int NT = NP / 16, R1 = W, R2 = NT - 1 - W;
    // f64 C/D layout of the 16x16x4 instruction: lane holds rows (lane >> 4) + 4 rg, column lane & 15
#pragma unroll
    for (int t = R1; t < NT; t++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = 16 * R1 + (lane >> 4) + 4 * rg, col = 16 * t + (lane & 15);
            if (row < nq && col < nq) Sp[(size_t)row * nq + col] = -acc[t - R1][rg];
        }
#pragma unroll
    for (int t = R2; t < NT; t++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = 16 * R2 + (lane >> 4) + 4 * rg, col = 16 * t + (lane & 15);
            if (row < nq && col < nq) Sp[(size_t)row * nq + col] = -acc[NT - R1 + t - R2][rg];
        }
}

// workgroup barrier that orders LDS traffic only: outstanding GLOBAL loads (the prefetch) stay in flight across it
__device__ __forceinline__ void ba_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8 wavefronts, specialised: wavefronts 0-3 CONSUME (matrix cores: NT + 1 tiles each of Z_g^T Z_g), wavefronts 4-7 PRODUCE (fetch the raw records two
// groups ahead, park them in LDS, turn them into Z_{g+1}, the reduced right-hand side and the diagonal blocks).  Every SIMD hosts one wavefront
// of each kind, so its matrix pipe and its vector pipe are fed by different instruction streams and really run at the same time -- a single
// wavefront issues in order and serialised the two (measured: MFMA, VALU and memory time simply added up).  One workgroup barrier per group.
// `pass` (NP = 256 only): consumer jobs 4 pass .. 4 pass + 3 of the NT / 2 = 8.  The diagonal blocks V_c and the right-hand side are accumulated and added by
// the LAST pass: by then every tile they touch is final (the earlier pass is a finished launch, this block's own tiles are stored before barrier (C)).
#define BA_SCHUR_THREADS 512
template <int NP, int PASS>
__global__ __launch_bounds__(BA_SCHUR_THREADS) void k_ba_schur_mfma(BaJob J)
{
    constexpr int pass = PASS;
    using Cfg = BaSchurCfg<NP>;
    constexpr int NH = Cfg::NH, NRW = Cfg::NRW, ND = Cfg::ND;
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = 6 * nc, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int pw = wave & 3;           // point slot (producer) / tile-row pair (consumer)
    const int ptid = tid & 255;        // thread index within its half
    const int RAWW = 20 * nc + 10, nwords = 20 * nc + 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sZ = reinterpret_cast<double*>(smem);   // [2][12][NP]
    double* sRaw = sZ + 2 * 12 * NP;                // [2][4][RAWW] raw records
    double* sR = sRaw + 2 * 4 * RAWW;               // [4][NP] rhs partials of the four producer waves
    const int chunk = (nt + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(nt, i0 + chunk);
    const long long nent = (long long)nq * nq;
    const bool last_pass = pass == Cfg::NT / 8 - 1;
    if (i0 >= i1) {  // the last workgroups of a launch can own no point (nt not a multiple of the chunk): their partials are zeros
        if (pass == 0) {
            double* Sp0 = J.Spart + (size_t)blockIdx.x * ((size_t)nq * nq);
            for (long long e = tid; e < (long long)nq * nq; e += BA_SCHUR_THREADS) Sp0[e] = 0.0;
            if (tid < nq) J.Rpart[(size_t)blockIdx.x * nq + tid] = 0.0;
        }
        return;
    }
    for (int q = tid; q < 2 * 12 * NP; q += BA_SCHUR_THREADS) sZ[q] = 0.0;  // zero padding of both buffers (columns >= nq stay zero)
    const int i_last = i1 - 1;
    double* Sp = J.Spart + (size_t)blockIdx.x * nent;

    if (producer) {
        double accD[ND];   // diagonal-block entries e = ptid + 256 k < nc * 36
        double accR[NH];   // rhs entries q = lane + 64 h of the points this wave handled
#pragma unroll
        for (int k = 0; k < ND; k++) accD[k] = 0.0;
#pragma unroll
        for (int h = 0; h < NH; h++) accR[h] = 0.0;
        BaRawOff<NRW> off;
        ba_raw_offsets<NRW>(off, J, lane);
        BaColOff<NH, ND> col;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int q = min(lane + 64 * h, nq - 1), c = q / 6, k = q - 6 * c;  // clamped: columns >= nq compute garbage that is stored as zeros
            col.jp[h] = 12 * nc + 6 * c; col.ju[h] = 12 * c + k; col.rr[h] = 18 * nc + 2 * c;
        }
#pragma unroll
        for (int k = 0; k < ND; k++) {
            const int e = min(ptid + 256 * k, nc * 36 - 1);  // clamped: the final store is guarded
            const int c = e / 36, rr = e - 36 * c, ka = rr / 6, kb = rr - 6 * ka;
            col.da[k] = 12 * c + ka; col.db[k] = 12 * c + kb;
        }
        // prologue: record and Z of group 0 (buffers 0)
        double Ra[NRW], Rb[NRW];
        ba_raw_fetch<NRW>(Ra, off, i0 + pw, i_last);
        ba_raw_fetch<NRW>(Rb, off, i0 + 4 + pw, i_last);
        __syncthreads();  // (A) sZ zero-fill complete
        ba_raw_park<NRW>(Ra, sRaw + pw * RAWW, lane, nwords, i0 + pw < i1);
        ba_raw_fetch<NRW>(Ra, off, i0 + 8 + pw, i_last);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own record is readable by this wave (LDS ops of a wave complete in order)
        ba_produce_z<NP>(accR, sZ, sRaw + pw * RAWW, col, nc, nq, lane, pw);
        ba_lds_barrier();  // (B) Z(0) and the records of group 0 complete
        // steady state, iteration g (consumers run MFMA(g) meanwhile): Rb = raw(g+1), Ra = raw(g+2); raw(g) lives in sRaw[buf]
        int buf = 0;
        for (int ig = i0; ig < i1; ig += 4, buf ^= 1) {
            double* recn = sRaw + ((buf ^ 1) * 4 + pw) * RAWW;
            if (!(J.dbg & 16)) ba_raw_park<NRW>(Rb, recn, lane, nwords, ig + 4 + pw < i1);
#pragma unroll
            for (int j = 0; j < NRW; j++) Rb[j] = Ra[j];
            if (!(J.dbg & 4)) ba_raw_fetch<NRW>(Ra, off, ig + 12 + pw, i_last);  // raw(g+3): two groups ahead (three measured no different)
            asm volatile("" ::: "memory");
            if (!(J.dbg & 2) && last_pass) ba_produce_diag<NH, ND>(accD, sRaw + buf * 4 * RAWW, RAWW, col);  // group g: all four records are complete since the last barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(J.dbg & 32)) ba_produce_z<NP>(accR, sZ + (buf ^ 1) * 12 * NP, recn, col, nc, nq, lane, pw);  // Z(g+1)
            if (!(J.dbg & 8)) ba_lds_barrier();  // Z(g+1) and records g+1 complete; consumers are done with Z(g)
        }
        // epilogue: after the consumers stored their tiles, add the diagonal blocks and write the rhs partial
#pragma unroll
        for (int h = 0; h < NH; h++) sR[pw * NP + lane + 64 * h] = accR[h];
        __threadfence_block();
        __syncthreads();  // (C)
        if (last_pass) {
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const int e = ptid + 256 * k;
                if (e < nc * 36) {
                    const int c = e / 36, rr = e - 36 * c, ka = rr / 6, kb = rr - 6 * ka;
                    // a 6x6 camera block can straddle two 16x16 tiles: entries with tile(row) > tile(col) are never read (k_ba_reduce mirrors the
                    // upper triangle), so only the others are stored
                    const int row = 6 * c + ka, col_ = 6 * c + kb;
                    if ((row >> 4) <= (col_ >> 4)) Sp[(size_t)row * nq + col_] += accD[k];
                }
            }
            for (int q = ptid; q < nq; q += 256) J.Rpart[(size_t)blockIdx.x * nq + q] = sR[q] + sR[NP + q] + sR[2 * NP + q] + sR[3 * NP + q];
        }
    } else {
        __syncthreads();   // (A)
        ba_lds_barrier();  // (B)
        // one loop per consumer job (not one loop with a switch inside): the accumulators of the four jobs then never meet in a phi node -- with the switch
        // inside the loop hipcc shuffled all NT + 1 tiles through v_mov_b64 every group and, at NP = 256, spilled six of them
        auto run_job = [&](auto wtag) {
            constexpr int W = decltype(wtag)::value;
            double4v acc[Cfg::NACC];
#pragma unroll
            for (int t = 0; t < Cfg::NACC; t++) acc[t] = double4v{0.0, 0.0, 0.0, 0.0};
            int buf = 0;
            for (int ig = i0; ig < i1; ig += 4, buf ^= 1) {
                if (!(J.dbg & 1)) ba_consume<NP, W>(acc, sZ + buf * 12 * NP, lane);
                if (!(J.dbg & 8)) ba_lds_barrier();
            }
            ba_syrk_store<NP, W>(acc, Sp, lane, nq);  // the upper-triangle tiles of -Z^T Z
        };
        switch (pw) {  // job 4 PASS + pw
        case 0: run_job(std::integral_constant<int, 4 * PASS + 0>{}); break;
        case 1: run_job(std::integral_constant<int, 4 * PASS + 1>{}); break;
        case 2: run_job(std::integral_constant<int, 4 * PASS + 2>{}); break;
        default: run_job(std::integral_constant<int, 4 * PASS + 3>{}); break;
        }
        __threadfence_block();
        __syncthreads();  // (C)
    }
}

// ---- Schur stage 1 on the matrix cores, 43..128 cameras (253..768 reduced unknowns) ------------------------------------------------
// Above 42 cameras a point's raw record (20 nc + 9 doubles) and a wavefront's share of the 16 x 16 tiles no longer fit LDS / the register file of
// k_ba_schur_mfma, and the VALU kernel k_ba_points re-forms W and Y once per 16 384 entries of S (3.8 ms per iteration at 50 cameras, 170 ms at 128:
// round 5).  Here the same contraction  S = V + I - sum_i Z_i^T Z_i  is split in two launches:
//   k_ba_zbuild    : Z = L^T W of every point, materialised ONCE (J.Y, [3 nt][6 nc] row-major, 92 MB at 128 cameras x 5000 points), the reduced
//                    right-hand side and the 6 x 6 diagonal blocks V_c (one partial per workgroup: J.Rpart / J.Dpart, summed by k_ba_reduce)
//   k_ba_syrk_mfma : a K-split symmetric rank-k update on v_mfma_f64_16x16x4_f64: workgroup (R <= C, split b) forms the 128 x 128 macro tile
//                    (R, C) of -Z_b^T Z_b over the rows of chunk b, 32 rows of both operand panels per LDS stage.  One workgroup per CU (the
//                    16 tiles of a wavefront are 128 accumulator registers), so the launch is sized to ONE resident round: npairs x nsplit <= 256.
//                    Z is read from L2 / the infinity cache 2 x (number of macro-tile columns) times.
// The partial systems are summed by k_ba_reduce exactly as for the smaller kernels (upper-triangle 16 x 16 tiles, mirrored there).
#define BA_ZB_PL 4  // points a k_ba_zbuild block works on at a time (threadIdx.y)
__global__ __launch_bounds__(256 * BA_ZB_PL) void k_ba_zbuild(BaJob J)
{
    ba_select_window(J, blockIdx.z);
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = 6 * nc, nf = nc + 1, tx = threadIdx.x, ty = threadIdx.y;
    const int chunk = (nt + gridDim.x - 1) / gridDim.x;  // (its own partition of the points: nothing to do with the K splits of k_ba_syrk_mfma)
    const int i0 = blockIdx.x * chunk, i1 = min(nt, i0 + chunk);
    // column q = 6 c + k of the reduced system (camera c + 1, parameter k)
    const int q = blockIdx.y * 256 + tx;
    const bool qon = q < nq;
    const int qc = min(q, nq - 1), c = qc / 6, k = qc - 6 * c;
    // entries of the diagonal blocks owned by this thread: e = e0 + tx + 256 j of this block's share of the 36 nc
    constexpr int ND = 6;
    const int nde = 36 * nc, per = (nde + (int)gridDim.y - 1) / (int)gridDim.y, e0 = blockIdx.y * per, e1 = min(nde, e0 + per);
    int da[ND], db[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) {
        const int e = min(e0 + tx + 256 * j, nde - 1), ce = e / 36, rr = e - 36 * ce, ka = rr / 6, kb = rr - 6 * ka;
        da[j] = 12 * (ce + 1) + ka; db[j] = 12 * (ce + 1) + kb;
    }
    double accR = 0.0, accD[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) accD[j] = 0.0;
    for (int i = i0 + ty; i < i1; i += BA_ZB_PL) {
        const size_t mp = (size_t)i * nf, m = mp + (c + 1);
        const double* jp = J.Jp + 6 * m;
        const double* jc = J.Jc + 12 * m;
        const double* Lp = J.Lc + 6 * (size_t)i;
        const double* tp = J.tp + 3 * (size_t)i;
        const double ju = jc[k], jv = jc[6 + k], ru = J.r[2 * m], rv = J.r[2 * m + 1];
        const double w0 = jp[0] * ju + jp[3] * jv, w1 = jp[1] * ju + jp[4] * jv, w2 = jp[2] * ju + jp[5] * jv;
        // L^T is upper triangular: rows (l00 l10 l20), (0 l11 l21), (0 0 l22); L is stored l00 l10 l11 l20 l21 l22
        const double z0 = Lp[0] * w0 + Lp[1] * w1 + Lp[3] * w2, z1 = Lp[2] * w1 + Lp[4] * w2, z2 = Lp[5] * w2;
        accR += ju * ru + jv * rv - (w0 * tp[0] + w1 * tp[1] + w2 * tp[2]);
        if (qon) {
            double* Z = J.Y + (size_t)(3 * i) * nq + q;
            Z[0] = z0; Z[nq] = z1; Z[2 * (size_t)nq] = z2;
        }
        const double* jcp = J.Jc + 12 * mp;
#pragma unroll
        for (int j = 0; j < ND; j++) accD[j] += jcp[da[j]] * jcp[db[j]] + jcp[da[j] + 6] * jcp[db[j] + 6];
    }
    // the BA_ZB_PL point lanes are combined in a fixed order
    __shared__ double sh[BA_ZB_PL][256];
    double* Dp = J.Dpart + (size_t)blockIdx.x * nde;
#pragma unroll
    for (int jj = 0; jj <= ND; jj++) {
        const int j = jj - 1;  // -1: the right-hand side
        __syncthreads();
        sh[ty][tx] = jj == 0 ? accR : accD[jj > 0 ? jj - 1 : 0];
        __syncthreads();
        if (ty == 0) {
            double t = sh[0][tx];
#pragma unroll
            for (int u = 1; u < BA_ZB_PL; u++) t += sh[u][tx];
            if (j < 0) {
                if (qon) J.Rpart[(size_t)blockIdx.x * nq + q] = t;
            } else {
                const int e = e0 + tx + 256 * j;
                if (e < e1) Dp[e] = t;  // entry (ka, kb) of V_c at 36 c + 6 ka + kb: k_ba_reduce adds the partials to the diagonal blocks
            }
        }
    }
}

#define BA_SY_KB 16  // rows of Z per LDS stage
__global__ __launch_bounds__(256, 2) void k_ba_syrk_mfma(BaJob J, int nm)
{
    ba_select_window(J, blockIdx.z);
    if (*J.done) return;
    const int nt = J.nt, nq = 6 * J.nc, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    // macro tile (R, C), R <= C, of the nm x nm grid of 128 x 128 tiles
    int p = blockIdx.x, R = 0;
    while (p >= nm - R) { p -= nm - R; R++; }
    const int Cm = R + p;
    const bool diag = R == Cm;
    const int chunk = (nt + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * chunk, i1 = min(nt, i0 + chunk);
    const int r0 = 3 * i0, r1 = 3 * i1;
    // LDS: [slab of 4 rows][16-column tile][lane = 16 (row & 3) + (column & 15)] -- an MFMA operand is one conflict-free 512-byte read
    __shared__ __attribute__((aligned(16))) double sA[BA_SY_KB / 4 * 8 * 64];
    __shared__ __attribute__((aligned(16))) double sB[BA_SY_KB / 4 * 8 * 64];
    // staging: wave w fills tiles 2 w, 2 w + 1 of every slab; lane -> (tile of the pair, row of the slab, column pair)
    const int st = 2 * wave + (lane >> 5), srow = (lane >> 3) & 3, scol = 16 * st + 2 * (lane & 7);
    const bool a_on = 128 * R + scol < nq, b_on = !diag && 128 * Cm + scol < nq;  // nq is even: a column pair is inside or outside
    const double* gA = J.Y + (size_t)(r0 + srow) * nq + 128 * R + scol;
    const double* gB = J.Y + (size_t)(r0 + srow) * nq + 128 * Cm + scol;
    const int lds_off = st * 64 + srow * 16 + 2 * (lane & 7);
    typedef double double2v __attribute__((ext_vector_type(2)));
    double2v pa[BA_SY_KB / 4], pb[BA_SY_KB / 4];
    auto fetch = [&](int kb) {  // rows r0 + kb .. + BA_SY_KB - 1 of both panels -> registers (zeros past the chunk / past column nq)
#pragma unroll
        for (int j = 0; j < BA_SY_KB / 4; j++) {
            const int row = r0 + kb + 4 * j + srow;
            const bool on = row < r1;
            const size_t off = (size_t)(kb + 4 * j) * nq;
            pa[j] = (on && a_on) ? *reinterpret_cast<const double2v*>(gA + off) : double2v{0.0, 0.0};
            if (!diag) pb[j] = (on && b_on) ? *reinterpret_cast<const double2v*>(gB + off) : double2v{0.0, 0.0};
        }
    };
    double4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4v{0.0, 0.0, 0.0, 0.0};
    const bool dead = diag && wr > wc;  // every tile of this wavefront lies below the diagonal
    const double* sBp = diag ? sA : sB;
    if (r0 < r1) fetch(0);
    for (int kb = 0; r0 + kb < r1; kb += BA_SY_KB) {
        __syncthreads();  // the previous stage has been read
#pragma unroll
        for (int j = 0; j < BA_SY_KB / 4; j++) {
            *reinterpret_cast<double2v*>(sA + j * 512 + lds_off) = pa[j];
            if (!diag) *reinterpret_cast<double2v*>(sB + j * 512 + lds_off) = pb[j];
        }
        __syncthreads();
        if (r0 + kb + BA_SY_KB < r1) fetch(kb + BA_SY_KB);  // in flight while the matrix cores work on this stage
        if (!dead) {
            // software pipeline: the operands of slab s + 1 are read before the 16 matrix-core instructions of slab s are issued (a wavefront blocks at
            // every MFMA issue, so reads placed after them expose their LDS round trip)
            double a[4], b[4], an[4], bn[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { a[t] = sA[(4 * wr + t) * 64 + lane]; b[t] = sBp[(4 * wc + t) * 64 + lane]; }
#pragma unroll
            for (int sl = 0; sl < BA_SY_KB / 4; sl++) {
                if (sl + 1 < BA_SY_KB / 4) {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        an[t] = sA[((sl + 1) * 8 + 4 * wr + t) * 64 + lane];
                        bn[t] = sBp[((sl + 1) * 8 + 4 * wc + t) * 64 + lane];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ta = 0; ta < 4; ta++)
#pragma unroll
                    for (int tb = 0; tb < 4; tb++) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; t++) { a[t] = an[t]; b[t] = bn[t]; }
            }
        }
    }
    if (dead) return;
    // f64 C/D layout of the 16x16x4 instruction: lane holds rows (lane >> 4) + 4 rg, column lane & 15
    double* Sp = J.Spart + (size_t)blockIdx.y * ((size_t)nq * nq);
#pragma unroll
    for (int ta = 0; ta < 4; ta++)
#pragma unroll
        for (int tb = 0; tb < 4; tb++) {
            const int tr = 8 * R + 4 * wr + ta, tc = 8 * Cm + 4 * wc + tb;
            if (tr > tc) continue;  // k_ba_reduce mirrors the upper-triangle tiles
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int row = 16 * tr + (lane >> 4) + 4 * rg, col = 16 * tc + (lane & 15);
                if (row < nq && col < nq) Sp[(size_t)row * nq + col] = -acc[ta][tb][rg];
            }
        }
}

// Schur stage 2a: sum the per-workgroup partials into the augmented system and add +I.  A block handles 64 consecutive
// entries; each of its 4 wavefronts sums a fixed slice of the partials (512-byte coalesced reads, 8 in flight), then the
// slices are combined in a fixed order (deterministic).
// nparts_r / ndpart (43+ cameras): partial right-hand sides and partial diagonal blocks V_c of the k_ba_zbuild workgroups (otherwise nparts and 0)
// NS = slices (wavefronts) per block: 4, or 16 when a window has many partials (one C5 window: 256 -- a wavefront then sums 16 of them in two rounds of
// eight loads in flight instead of 64 in eight rounds: the kernel is bound by those dependent round trips, not by its 19 MB)
template <int NS>
__global__ __launch_bounds__(64 * NS) void k_ba_reduce(BaJob J, int nparts, int nparts_r, int ndpart)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nq = J.nq, ld = nq + 1;
    const long long nent = (long long)nq * nq, ntot = nent + nq;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    __shared__ double sh[NS][64];
    double s = 0.0;
    // zmode: only the tiles with tile(row) <= tile(col) were written; the owners of those entries also write the mirrored one, the
    // entries below do nothing (reading the partials transposed instead cost 3.5x the kernel)
    bool lower = false;
    if (J.zmode && e < nent) {
        const int a = (int)(e / nq), b = (int)(e - (long long)a * nq);
        lower = (a >> 4) > (b >> 4);
    }
    if (e < ntot && !lower) {
        const int np = e < nent ? nparts : nparts_r;
        const int per = (np + NS - 1) / NS, p0 = slice * per, p1 = min(np, p0 + per);
        const double* src = e < nent ? J.Spart + e : J.Rpart + (e - nent);
        const size_t step = e < nent ? (size_t)nent : (size_t)nq;
        if (ndpart && e < nent) {  // an entry of a diagonal block: + the partials of V_c
            const int a = (int)(e / nq), b = (int)(e - (long long)a * nq), ca = a / 6;
            if (ca == b / 6) {
                const double* d = J.Dpart + 36 * ca + 6 * (a - 6 * ca) + (b - 6 * ca);
                const int dper = (ndpart + NS - 1) / NS, d0 = slice * dper, d1 = min(ndpart, d0 + dper);
                for (int q = d0; q < d1; q++) s += d[(size_t)q * (36 * (size_t)J.nc)];
            }
        }
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)(p + u) * step];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; p < p1; p++) s += src[(size_t)p * step];
    }
    sh[slice][lane] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // fold the residual partials of k_ba_jac (previous launch) into acc[0]
        double t = 0.0;
        for (int k = 0; k < 16; k++) { t += J.rslot[k]; J.rslot[k] = 0.0; }
        J.acc[0] += t;
    }
    __syncthreads();
    if (slice == 0 && e < ntot && !lower) {
        double t = 0.0;
        for (int k = 0; k < NS; k++) t += sh[k][lane];
        if (e < nent) {
            const int a = (int)(e / nq), b = (int)(e - (long long)a * nq);
            J.Sfull[(size_t)a * ld + b] = t + ((a == b && J.add_identity) ? 1.0 : 0.0);  // sharded runs: rank 0 adds the +I
            if (J.zmode && (a >> 4) < (b >> 4)) J.Sfull[(size_t)b * ld + a] = t;
        } else {
            J.Sfull[(size_t)(e - nent) * ld + nq] = t;
        }
    }
}

// Schur stage 2b: solve S dc = rhs  (the reference: inv(JtJ + I) @ Jt r, NLS.py:235).
// S = V + I - W^T (U + I)^-1 W is the Schur complement of the symmetric positive definite JtJ + I, hence itself SPD:
// Gaussian elimination needs no pivoting there (it is backward stable for SPD systems; every leading diagonal block is
// SPD too), so the result equals LAPACK's partially pivoted inverse to rounding.
// The augmented (6nc) x (6nc+1) system is REGISTER resident: thread (rr, kk) of a G x G arrangement owns rows rr + G m
// (m < RM) and columns kk + G j (j < CM).  The single workgroup runs on one CU and is bound by the dependent-latency chain
// of an elimination round (barrier, LDS round trip, reciprocal, FMAs), not by work -- so a round eliminates a 2 x 2 pivot
// block: the owners of rows / columns c, c+1 have published them (double-buffered LDS vectors), every thread forms its
// multipliers F = [A(r,c) A(r,c+1)] P^-1, applies the rank-2 update to its own elements and the owners of the next block
// publish right away.  One barrier per TWO unknowns; a last scalar round handles an odd count.  (History: LDS-resident
// pivoted Gauss-Jordan 200-250 us at nc = 19 -> register-resident scalar rounds 80 us -> this.)
template <int RM, int CM, int G>
__device__ __forceinline__ void ba_publish(const double (&a)[RM][CM], double* row_slot, double* col_slot, int rr, int kk, int jn, int pos)
{
    // row / column with position `pos` inside group jn (rows and columns use the same group width, so the row group is jn too).
    // jn is a constant after the caller's outer loop is unrolled: the register indices below are static.
    if (rr == pos) {
#pragma unroll
        for (int j = 0; j < CM; j++)
            if (j >= jn) row_slot[kk + G * j] = a[jn < RM ? jn : 0][j];
    }
    if (kk == pos) {
#pragma unroll
        for (int m = 0; m < RM; m++) col_slot[rr + G * m] = a[m][jn < CM ? jn : 0];
    }
}

template <int RM, int CM, int G>
__global__ __launch_bounds__(G * G) void k_ba_solve(BaJob J, int nparts)
{
    ba_select_window(J, blockIdx.y);
    (void)nparts;
    if (*J.done) return;
    static_assert(G % 2 == 0, "a pivot pair never straddles a group");
    constexpr int NTH = G * G;
    const int nq = J.nq, tid = threadIdx.x, ld = nq + 1;
    const int rr = tid / G, kk = tid % G;
    __shared__ double s_row[2][2][G * CM];   // [buffer][row of the pair][column]
    __shared__ double s_colv[2][2][G * RM];  // [buffer][column of the pair][row]
    __shared__ double s_pinv[G * RM / 2 + 1][4];  // inverse pivot blocks (a scalar round stores {1/pivot, 0, 0, 0})
    // the partials were reduced into Sfull ([nq][nq+1], +I included) by k_ba_reduce
    double a[RM][CM];
#pragma unroll
    for (int m = 0; m < RM; m++)
#pragma unroll
        for (int j = 0; j < CM; j++) {
            const int r = rr + G * m, k = kk + G * j;
            a[m][j] = (r < nq && k <= nq) ? J.Sfull[(size_t)r * ld + k] : 0.0;
        }
    ba_publish<RM, CM, G>(a, s_row[0][0], s_colv[0][0], rr, kk, 0, 0);
    ba_publish<RM, CM, G>(a, s_row[0][1], s_colv[0][1], rr, kk, 0, 1);
    __syncthreads();
    // the column-group index of the pivots is a compile-time constant inside the unrolled outer loop: no dynamic register
    // indexing, finished column groups drop out of the update statically
#pragma unroll
    for (int cj = 0; cj < CM; cj++) {
        for (int ck = 0; ck < G; ck += 2) {
            const int c = G * cj + ck;
            if (c >= nq) break;
            const int buf = (c >> 1) & 1;
            const bool pair = c + 1 < nq;
            double f0[RM], f1[RM];
            if (pair) {
                const double p00 = s_colv[buf][0][c], p01 = s_colv[buf][1][c], p10 = s_colv[buf][0][c + 1], p11 = s_colv[buf][1][c + 1];
                const double inv = 1.0 / (p00 * p11 - p01 * p10);
                const double i00 = p11 * inv, i01 = -p01 * inv, i10 = -p10 * inv, i11 = p00 * inv;
                if (tid == 0) { s_pinv[c >> 1][0] = i00; s_pinv[c >> 1][1] = i01; s_pinv[c >> 1][2] = i10; s_pinv[c >> 1][3] = i11; }
#pragma unroll
                for (int m = 0; m < RM; m++) {
                    const int r = rr + G * m;
                    const double v0 = s_colv[buf][0][r], v1 = s_colv[buf][1][r];
                    const bool piv = (r == c) || (r == c + 1);  // the pivot rows stay
                    f0[m] = piv ? 0.0 : v0 * i00 + v1 * i10;
                    f1[m] = piv ? 0.0 : v0 * i01 + v1 * i11;
                }
            } else {  // last unknown of an odd system: scalar round
                const double inv = 1.0 / s_colv[buf][0][c];
                if (tid == 0) { s_pinv[c >> 1][0] = inv; s_pinv[c >> 1][1] = 0.0; s_pinv[c >> 1][2] = 0.0; s_pinv[c >> 1][3] = 0.0; }
#pragma unroll
                for (int m = 0; m < RM; m++) {
                    f0[m] = (rr + G * m == c) ? 0.0 : s_colv[buf][0][rr + G * m] * inv;
                    f1[m] = 0.0;
                }
            }
#pragma unroll
            for (int j = cj; j < CM; j++) {
                // column group cj: only the columns behind the pair; later groups: all (columns past nq hold zeros)
                if (j > cj || kk > ck + (pair ? 1 : 0)) {
                    const double r0 = s_row[buf][0][kk + G * j], r1 = pair ? s_row[buf][1][kk + G * j] : 0.0;
#pragma unroll
                    for (int m = 0; m < RM; m++) a[m][j] = __builtin_fma(-f1[m], r1, __builtin_fma(-f0[m], r0, a[m][j]));
                }
            }
            // publish the next pair (the other buffer: slow wavefronts may still read this one)
            if (c + 2 < nq) {
                if (ck + 2 < G) {
                    ba_publish<RM, CM, G>(a, s_row[buf ^ 1][0], s_colv[buf ^ 1][0], rr, kk, cj, ck + 2);
                    if (c + 3 < nq) ba_publish<RM, CM, G>(a, s_row[buf ^ 1][1], s_colv[buf ^ 1][1], rr, kk, cj, ck + 3);
                } else if (cj + 1 < CM) {
                    ba_publish<RM, CM, G>(a, s_row[buf ^ 1][0], s_colv[buf ^ 1][0], rr, kk, cj + 1, 0);
                    if (c + 3 < nq) ba_publish<RM, CM, G>(a, s_row[buf ^ 1][1], s_colv[buf ^ 1][1], rr, kk, cj + 1, 1);
                }
            }
            __syncthreads();
        }
    }
    // dc = P^-1 rhs per pivot block
    if (kk == (nq % G)) {
#pragma unroll
        for (int j = 0; j < CM; j++)
            if (j == (nq / G)) {
#pragma unroll
                for (int m = 0; m < RM; m++) s_colv[0][0][rr + G * m] = a[m][j];
            }
    }
    __syncthreads();
    for (int q = tid; q < nq; q += NTH) {
        const int b = q >> 1, q0 = 2 * b;
        const double r0 = s_colv[0][0][q0], r1 = (q0 + 1 < nq) ? s_colv[0][0][q0 + 1] : 0.0;
        J.dc[q] = (q & 1) ? s_pinv[b][2] * r0 + s_pinv[b][3] * r1 : s_pinv[b][0] * r0 + s_pinv[b][1] * r1;
    }
}

// Schur stage 2b on the matrix cores (nq <= 124): BLOCK Gauss-Jordan with 4 x 4 pivot blocks, the augmented system resident in the accumulator
// registers of v_mfma_f64_16x16x4_f64.  The 128 x 128 padded matrix M = [S | 0 | rhs] (rhs in column 127, identity on the padded diagonal) is an
// 8 x 8 grid of 16 x 16 tiles; wavefront w of 4 owns tile rows w and w + 4 (16 tiles = 128 accumulator registers).  Round r eliminates columns
// c0 = 4r .. 4r+3 from every row but the pivot rows:  M -= F R  with R = the 4 pivot rows (4 x 128) and F = M[:, c0:c0+4] P^-1 (128 x 4, zero in
// the pivot rows) -- exactly the rank-4 update the instruction performs per tile (A = -F tile rows, B = R tile columns).  Tile columns left of
// the pivot are finished (their pivot-row entries are zero) and are skipped statically.  Per round: one barrier, the pivot rows / columns
// travel through double-buffered LDS vectors, every lane inverts the 4 x 4 pivot block itself (two dependent reciprocals), 2 x (8 - r/4) MFMAs.
// 29 rounds at nq = 114 instead of the 57 rank-2 rounds of the VALU kernel below, and the 16 K multiply-adds of a round are 16 instructions
// per wavefront instead of 128 v_fma_f64: 64 us -> see DESIGN.md section 6.  S is SPD, so no pivoting is needed (every leading block is SPD).
// 1 / x from v_rcp_f64 + two Newton steps (relative error ~1e-16; the IEEE division sequence is 11 dependent instructions, this is 5)
__device__ __forceinline__ double ba_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}

// column `col` (0..3, per lane) of the inverse of the 4 x 4 pivot block P, through block LU on 2 x 2 blocks:
//   P = [[A B],[C D]], T = A^-1 B, Sc = D - C T, V = Sc^-1 C A^-1;  P^-1 = [[A^-1 + T V, -T Sc^-1],[-V, Sc^-1]]
// Two dependent reciprocals; the shared intermediates are the same in every lane, only the last four products depend on the column.
__device__ __forceinline__ void ba_inv4_col(const double (&P)[4][4], int col, double (&q)[4])
{
    const double ia = ba_rcp(__builtin_fma(P[0][0], P[1][1], -(P[0][1] * P[1][0])));
    const double a00 = P[1][1] * ia, a01 = -P[0][1] * ia, a10 = -P[1][0] * ia, a11 = P[0][0] * ia;
    const double t00 = __builtin_fma(a00, P[0][2], a01 * P[1][2]), t01 = __builtin_fma(a00, P[0][3], a01 * P[1][3]);
    const double t10 = __builtin_fma(a10, P[0][2], a11 * P[1][2]), t11 = __builtin_fma(a10, P[0][3], a11 * P[1][3]);
    const double s00 = __builtin_fma(-P[2][1], t10, __builtin_fma(-P[2][0], t00, P[2][2])), s01 = __builtin_fma(-P[2][1], t11, __builtin_fma(-P[2][0], t01, P[2][3]));
    const double s10 = __builtin_fma(-P[3][1], t10, __builtin_fma(-P[3][0], t00, P[3][2])), s11 = __builtin_fma(-P[3][1], t11, __builtin_fma(-P[3][0], t01, P[3][3]));
    const double is = ba_rcp(__builtin_fma(s00, s11, -(s01 * s10)));
    const double d00 = s11 * is, d01 = -s01 * is, d10 = -s10 * is, d11 = s00 * is;  // Sc^-1
    const double u00 = __builtin_fma(P[2][0], a00, P[2][1] * a10), u01 = __builtin_fma(P[2][0], a01, P[2][1] * a11);  // U = C A^-1
    const double u10 = __builtin_fma(P[3][0], a00, P[3][1] * a10), u11 = __builtin_fma(P[3][0], a01, P[3][1] * a11);
    const double v00 = __builtin_fma(d00, u00, d01 * u10), v01 = __builtin_fma(d00, u01, d01 * u11);                  // V = Sc^-1 U
    const double v10 = __builtin_fma(d10, u00, d11 * u10), v11 = __builtin_fma(d10, u01, d11 * u11);
    // column c < 2: (a_.c + T V_.c ; -V_.c)      column c >= 2: (-T D_.c' ; D_.c')  with c' = c - 2
    const bool left = col < 2, odd = (col & 1) != 0;
    const double g0 = left ? (odd ? v01 : v00) : (odd ? d01 : d00), g1 = left ? (odd ? v11 : v10) : (odd ? d11 : d10);
    const double h0 = left ? (odd ? a01 : a00) : 0.0, h1 = left ? (odd ? a11 : a10) : 0.0;
    const double e0 = __builtin_fma(t00, g0, t01 * g1), e1 = __builtin_fma(t10, g0, t11 * g1);
    q[0] = left ? h0 + e0 : -e0;
    q[1] = left ? h1 + e1 : -e1;
    q[2] = left ? -g0 : g0;
    q[3] = left ? -g1 : g1;
}

#define BA_GJ_N 128
#define BA_GJ_MAXQ 124  // the last pivot block must end before the rhs column (127)
#define BA_GJ_WAVES 8   // one 16-row tile row per wavefront, two wavefronts per SIMD: the per-round hand-over (accumulator read-back, LDS publish, waits) is
                        // per-wavefront serial work, so halving each wavefront's share and letting two of them interleave on a SIMD shortens the round
__global__ __launch_bounds__(64 * BA_GJ_WAVES) void k_ba_solve_mfma(BaJob J)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nq = J.nq, ld = nq + 1, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // tile row of this wavefront
    const int lr = lane >> 4, lc = lane & 15;
    __shared__ double s_R[2][4][BA_GJ_N];       // pivot rows of the coming round
    __shared__ double s_C[2][4][BA_GJ_N];       // pivot columns of the coming round
    __shared__ double s_pinv[BA_GJ_N / 4][16];  // inverse pivot blocks
    __shared__ double s_rhs[BA_GJ_N];
    double4v acc[8];
#pragma unroll
    for (int Jt = 0; Jt < 8; Jt++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = 16 * w + lr + 4 * rg, col = 16 * Jt + lc;
            double v;
            if (row < nq) v = col < nq ? J.Sfull[(size_t)row * ld + col] : (col == BA_GJ_N - 1 ? J.Sfull[(size_t)row * ld + nq] : 0.0);
            else v = (row == col && col != BA_GJ_N - 1) ? 1.0 : 0.0;
            acc[Jt][rg] = v;
        }
    // The wavefront that holds the coming round's pivot rows publishes them, reads the 4 x 4 pivot block back (its own LDS writes complete in order) and
    // publishes the block's INVERSE too: the other seven wavefronts then read their column of it (4 doubles) instead of the block (16) and skip the
    // reciprocal chain -- the round's LDS read phase is throughput bound (8 wavefronts x 14 KB), and the owner's extra work overlaps the others' wait.
    auto publish_inverse = [&](int buf_, int rn_) {  // owner wavefront only: s_R[buf_][.][4 rn_ ..] was just written by this very wavefront
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double P[4][4], pk[4];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 4; k++) P[m][k] = s_R[buf_][m][4 * rn_ + k];
        ba_inv4_col(P, lr, pk);
        if (lc == 0) {
#pragma unroll
            for (int m = 0; m < 4; m++) s_pinv[rn_][4 * m + lr] = pk[m];
        }
    };
    // publish the pivot rows / columns (and the inverse pivot block) of round 0
    if (w == 0) {
#pragma unroll
        for (int Jt = 0; Jt < 8; Jt++) s_R[0][lr][16 * Jt + lc] = acc[Jt][0];
        publish_inverse(0, 0);
    }
    if (lc < 4) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) s_C[0][lc][16 * w + lr + 4 * rg] = acc[0][rg];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BA_GJ_MAXQ / 4; r++) {
        const int c0 = 4 * r;
        if (c0 >= nq) break;
        const int buf = r & 1, J0 = r >> 2;
        // my column lr of P^-1 (the A operand of lane l is F[16 I + (l & 15)][l >> 4] = sum_m C[.][m] Pinv[m][l >> 4]), pivot columns of my rows,
        // pivot rows of my tile columns
        double pk[4], cv[4], b[8];
#pragma unroll
        for (int m = 0; m < 4; m++) pk[m] = s_pinv[r][4 * m + lr];
#pragma unroll
        for (int m = 0; m < 4; m++) cv[m] = s_C[buf][m][16 * w + lc];
#pragma unroll
        for (int Jt = 0; Jt < 8; Jt++)
            if (Jt >= J0) b[Jt] = s_R[buf][lr][16 * Jt + lc];
        const int i = 16 * w + lc;
        const double fv = cv[0] * pk[0] + cv[1] * pk[1] + cv[2] * pk[2] + cv[3] * pk[3];
        const double f = (i >= c0 && i < c0 + 4) ? 0.0 : -fv;  // the pivot rows stay
        const bool has_next = c0 + 4 < nq;
        const int rn = r + 1, In = rn >> 2, rgn = rn & 3, Jn = rn >> 2, cmn = 4 * (rn & 3);
        // the tile column of the next pivot first: its results are published while the other tiles are still in the matrix pipe
        if (has_next) acc[Jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(f, b[Jn], acc[Jn], 0, 0, 0);
#pragma unroll
        for (int Jt = 0; Jt < 8; Jt++)
            if (Jt >= J0 && !(has_next && Jt == Jn)) acc[Jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(f, b[Jt], acc[Jt], 0, 0, 0);
        // publish the next pivot rows / columns (the other buffer: slow wavefronts may still read this one)
        if (has_next) {
            if (lc >= cmn && lc < cmn + 4) {
#pragma unroll
                for (int rg = 0; rg < 4; rg++) s_C[buf ^ 1][lc - cmn][16 * w + lr + 4 * rg] = acc[Jn][rg];
            }
            if (w == In) {
#pragma unroll
                for (int Jt = 0; Jt < 8; Jt++)
                    if (Jt >= Jn) s_R[buf ^ 1][lr][16 * Jt + lc] = acc[Jt][rgn];
                publish_inverse(buf ^ 1, rn);
            }
        }
        __syncthreads();
    }
    // the right-hand side column (127 = tile column 7, local column 15), then dc = P_r^-1 rhs_r per pivot block
    if (lc == 15) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) s_rhs[16 * w + lr + 4 * rg] = acc[7][rg];
    }
    __syncthreads();
    if (tid < nq) {
        const int r = tid >> 2, k = tid & 3;
        const double* Qr = s_pinv[r];
        J.dc[tid] = Qr[4 * k] * s_rhs[4 * r] + Qr[4 * k + 1] * s_rhs[4 * r + 1] + Qr[4 * k + 2] * s_rhs[4 * r + 2] + Qr[4 * k + 3] * s_rhs[4 * r + 3];
    }
}

// Schur stage 2b for more than 124 unknowns (21+ cameras): BLOCKED CHOLESKY across launches.  The augmented system of 125..768 unknowns does not fit the
// register file of one CU (252 x 253 doubles = 510 KB of its 512 KB), which is what made the register-resident Gauss-Jordan of round 2 spill (620 VGPRs,
// 1.3 ms per solve at 36-42 cameras) and the in-L2 elimination crawl (5 ms at 50).  S is SPD (Schur complement of J^T J + I), so S = L L^T, right-looking,
// panels of 32 columns, in place in the lower triangle of Sfull (it stays in L2); the right-hand side rides along as one more row (stored where it already
// is: column nq), so when the last panel is done it holds y = L^-1 rhs.  Per panel two launches:
//   k_ba_chol_panel : ONE workgroup -- the 32 x 32 diagonal block factored in LDS, then every row below it (one thread per row, the rhs row included)
//                     forward-substituted against it (32 registers per row)
//   k_ba_chol_update: the trailing lower triangle (and the rhs row) minus the panel's outer product, one workgroup per 32 x 32 tile
// and at the end k_ba_chol_back: L^T dc = y, panel by panel from the last (a matrix-vector product over the rows below, then a 32-step triangular solve).
// Measured at 252 unknowns (8 panels): panel kernel 17-32 us (4.6 us the register Cholesky of the diagonal block, up to 9 us the rows below, the rest
// launch + three dependent L2 round trips), update 6.7 us, back-substitution 28 us: 265 us per solve against 1331 us (the spilling elimination kernel).  Same result as the elimination kernels to rounding (no pivoting needed or done in either).
#define BA_CH_NB 32
// wave-uniform copy of lane `l`'s double (l is a constant after unrolling: two v_readlane_b32 with an immediate lane)
__device__ __forceinline__ double ba_readlane_f64(double v, int l)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// 1 / sqrt(x): v_rsq_f64 + two Newton steps (relative error ~1e-16)
__device__ __forceinline__ double ba_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
}
// Cholesky of a 32 x 32 SPD block held one ROW per lane (lanes 0..31 of one wavefront, a[j] = entry (lane, j), upper part zero): right-looking, the
// pivot and the column below it travel by v_readlane (constant lane numbers after unrolling: the whole factorisation is straight-line code, no LDS, no
// barrier).  The LDS version with two workgroup barriers per column took 1 us per column (35 us per panel kernel).
__device__ __forceinline__ void ba_chol32_rows(double (&a)[BA_CH_NB])
{
#pragma unroll
    for (int k = 0; k < BA_CH_NB; k++) {
        const double inv = ba_rsqrt(ba_readlane_f64(a[k], k));
        a[k] = a[k] * inv;  // L(i, k) for the lanes below the pivot; the pivot lane gets sqrt(d); lanes above hold zeros
#pragma unroll
        for (int j = k + 1; j < BA_CH_NB; j++) a[j] = __builtin_fma(-a[k], ba_readlane_f64(a[k], j), a[j]);
    }
}
__global__ __launch_bounds__(256) void k_ba_chol_panel(BaJob J, int k0)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nq = J.nq, ld = nq + 1, tid = threadIdx.x;
    const int nb = min(BA_CH_NB, nq - k0);
    double* A = J.Sfull;
    __shared__ double sD[BA_CH_NB][BA_CH_NB + 1];
    __shared__ double sDt[BA_CH_NB][BA_CH_NB];  // transposed factor: sDt[k][j] = L_D[j][k] -- the column a forward-substitution step needs is one contiguous row
    __shared__ double sV[256][BA_CH_NB + 1];
    __shared__ double sDi[BA_CH_NB];
    const int mrows = nq - k0 - nb;  // rows below the block; index mrows = the rhs row
    // slab of up to 256 rows below the block (+ the rhs row) into LDS, coalesced (consecutive threads = consecutive columns of a row; a thread walking its own
    // row touched 64 cache lines per load)
    auto load_slab = [&](int t0, int cnt) {
        for (int e = tid; e < cnt * BA_CH_NB; e += 256) {
            const int rr = e / BA_CH_NB, j = e - rr * BA_CH_NB, t = t0 + rr;
            double v = 0.0;
            if (j < nb) v = t == mrows ? A[(size_t)(k0 + j) * ld + nq] : A[(size_t)(k0 + nb + t) * ld + k0 + j];
            sV[rr][j] = v;
        }
    };
    for (int e = tid; e < BA_CH_NB * BA_CH_NB; e += 256) {
        const int i = e / BA_CH_NB, j = e - i * BA_CH_NB;
        sD[i][j] = (i < nb && j <= i) ? A[(size_t)(k0 + i) * ld + k0 + j] : (i == j ? 1.0 : 0.0);  // (a short last panel is padded with the identity)
    }
    load_slab(0, min(256, mrows + 1));  // does not depend on the factorisation: its round trip overlaps the diagonal block's
    __syncthreads();
    // the diagonal block is factored by ONE wavefront in registers (a row per lane), the reciprocals of its diagonal are kept for the rows below
    if (tid < 64) {
        const int i = tid & (BA_CH_NB - 1);  // (lanes 32..63 duplicate lanes 0..31; only the lower 32 are read through v_readlane)
        double a[BA_CH_NB];
#pragma unroll
        for (int j = 0; j < BA_CH_NB; j++) a[j] = sD[i][j];
        ba_chol32_rows(a);
        if (tid < BA_CH_NB) {
#pragma unroll
            for (int j = 0; j < BA_CH_NB; j++) { sD[i][j] = a[j]; sDt[j][i] = a[j]; }
            double d = 0.0;
#pragma unroll
            for (int j = 0; j < BA_CH_NB; j++) d = j == i ? a[j] : d;
            sDi[i] = 1.0 / d;
        }
    }
    __syncthreads();
    // the factor of the diagonal block goes back (lower triangle)
    for (int e = tid; e < BA_CH_NB * BA_CH_NB; e += 256) {
        const int i = e / BA_CH_NB, j = e - i * BA_CH_NB;
        if (i < nb && j <= i) A[(size_t)(k0 + i) * ld + k0 + j] = sD[i][j];
    }
    // rows below the block + the rhs row: v <- v L_D^-T, one thread per row, RIGHT-looking (v[k] is final, then every later entry drops its term: the 31 - k
    // updates of a step are independent and their operands -- row k of the transposed factor -- are read in one go; the left-looking form waited for LDS
    // after nearly every multiply-add: 270 waits, 7 us)
    for (int t0 = 0; t0 <= mrows; t0 += 256) {
        const int cnt = min(256, mrows + 1 - t0);
        if (t0 > 0) {
            __syncthreads();  // (the previous slab's stores have been issued from sV)
            load_slab(t0, cnt);
            __syncthreads();
        }
        if (tid < cnt) {
            double v[BA_CH_NB];
#pragma unroll
            for (int j = 0; j < BA_CH_NB; j++) v[j] = sV[tid][j];
#pragma unroll
            for (int k = 0; k < BA_CH_NB; k++) {
                v[k] = v[k] * sDi[k];
#pragma unroll
                for (int j = k + 1; j < BA_CH_NB; j++) v[j] = __builtin_fma(-v[k], sDt[k][j], v[j]);
            }
#pragma unroll
            for (int j = 0; j < BA_CH_NB; j++) sV[tid][j] = v[j];
        }
        __syncthreads();
        for (int e = tid; e < cnt * BA_CH_NB; e += 256) {
            const int rr = e / BA_CH_NB, j = e - rr * BA_CH_NB, t = t0 + rr;
            if (j < nb) {
                if (t == mrows) A[(size_t)(k0 + j) * ld + nq] = sV[rr][j];
                else A[(size_t)(k0 + nb + t) * ld + k0 + j] = sV[rr][j];
            }
        }
    }
}

// trailing update after panel k0: A[i][j] -= sum_k L[i][k0+k] L[j][k0+k] for the lower-triangle tile (blockIdx.x >= blockIdx.y) of 32 x 32 behind the panel;
// tile row index == number of row tiles means the rhs row (y[j] -= sum_k yp[k] L[j][k0+k])
__global__ __launch_bounds__(256) void k_ba_chol_update(BaJob J, int k0)
{
    ba_select_window(J, blockIdx.z);
    if (*J.done) return;
    const int nq = J.nq, ld = nq + 1, tid = threadIdx.x;
    const int nb = min(BA_CH_NB, nq - k0), r0 = k0 + nb;
    const int ntile = (nq - r0 + BA_CH_NB - 1) / BA_CH_NB;
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj >= ntile || ti > ntile || ti < tj) return;
    double* A = J.Sfull;
    const bool rhs = ti == ntile;
    __shared__ double sLi[BA_CH_NB][BA_CH_NB + 1], sLj[BA_CH_NB][BA_CH_NB + 1];
    for (int e = tid; e < BA_CH_NB * BA_CH_NB; e += 256) {
        const int i = e / BA_CH_NB, k = e - i * BA_CH_NB;
        const int gi = r0 + ti * BA_CH_NB + i, gj = r0 + tj * BA_CH_NB + i;
        double li = 0.0;
        if (k < nb) {
            if (rhs) li = i == 0 ? A[(size_t)(k0 + k) * ld + nq] : 0.0;  // the rhs row's panel entries (already solved): one row
            else if (gi < nq) li = A[(size_t)gi * ld + k0 + k];
        }
        sLi[i][k] = li;
        sLj[i][k] = (k < nb && gj < nq) ? A[(size_t)gj * ld + k0 + k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int e = tid + 256 * u, i = e / BA_CH_NB, j = e - i * BA_CH_NB;
        const int gi = r0 + ti * BA_CH_NB + i, gj = r0 + tj * BA_CH_NB + j;
        if (gj >= nq) continue;
        if (rhs) {
            if (i != 0) continue;
        } else if (gi >= nq || gj > gi) continue;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < BA_CH_NB; k++) acc = __builtin_fma(sLi[i][k], sLj[j][k], acc);
        double* dst = rhs ? A + (size_t)gj * ld + nq : A + (size_t)gi * ld + gj;
        *dst -= acc;
    }
}

// LEFT-looking panel step (round 5): ONE launch per panel instead of two, and the rows below the diagonal block are spread over workgroups instead of walked
// by one.  Workgroup b owns BA_CL_RB rows below the block (the rhs row is the last one).  Nothing right of column k0 has been touched yet, so it first
// brings the panel's columns of its rows up to date -- C = A[rows, k0:k0+32] - L[rows, 0:k0] L[k0:k0+32, 0:k0]^T, the same for the 32 diagonal rows (every
// workgroup forms the diagonal block redundantly: 32 x 32 x k0 flops buy the absence of a grid-wide dependency) -- then factors the diagonal block (one
// wavefront, registers, as above) and forward-substitutes its rows.  Only workgroup 0 writes the diagonal factor back.
// At 768 unknowns (24 panels): 1115 + 175 us for k_ba_chol_panel + k_ba_chol_update per solve before.
#define BA_CL_RB 32
#define BA_CL_KC 128                          // columns of L per LDS stage
#define BA_CL_XP (BA_CL_KC + 2)               // pitch = 4 banks mod 64: the 16 rows x 4 columns of an MFMA operand read hit 64 different banks
#define BA_CL_LDS ((BA_CH_NB + BA_CL_RB) * BA_CL_XP * 8)
__global__ __launch_bounds__(256) void k_ba_chol_left(BaJob J, int k0)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nq = J.nq, ld = nq + 1, tid = threadIdx.x;
    const int nb = min(BA_CH_NB, nq - k0), r0 = k0 + nb;
    const int mrows = nq - r0;  // rows below the block; index mrows = the rhs row
    const int t0 = blockIdx.x * BA_CL_RB, cnt = min(BA_CL_RB, mrows + 1 - t0);
    double* A = J.Sfull;
    constexpr int KC = BA_CL_KC, XP = BA_CL_XP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double (*sX)[XP] = reinterpret_cast<double (*)[XP]>(smem);  // [64][XP]: rows 0..31 the diagonal rows, 32.. this workgroup's rows
    // (after the products the same memory holds the diagonal block, its transposed factor, this workgroup's rows and the reciprocal diagonal)
    double (*sD)[BA_CH_NB + 1] = reinterpret_cast<double (*)[BA_CH_NB + 1]>(smem);
    double (*sDt)[BA_CH_NB] = reinterpret_cast<double (*)[BA_CH_NB]>(smem + sizeof(double) * BA_CH_NB * (BA_CH_NB + 1));
    double (*sV)[BA_CH_NB + 1] = reinterpret_cast<double (*)[BA_CH_NB + 1]>(smem + sizeof(double) * (BA_CH_NB * (BA_CH_NB + 1) + BA_CH_NB * BA_CH_NB));
    double* sDi = reinterpret_cast<double*>(smem + sizeof(double) * (BA_CH_NB * (BA_CH_NB + 1) + BA_CH_NB * BA_CH_NB + BA_CL_RB * (BA_CH_NB + 1)));
    // staging: thread -> 32 elements of a 64 x 128 chunk; wavefront w reads rows w, w + 4, ... -- one row (2 x 512 contiguous bytes) per pair of load
    // instructions, so a row's address is wave uniform (scalar base + one shared lane offset: no vector arithmetic per load).  The next chunk is in
    // flight while the matrix cores work on this one: the panel's columns come from the infinity cache / HBM (written by other XCDs in the previous
    // launch), ~1.7 us per round trip, against ~2 us of matrix-core time per chunk.
    // Row r of a chunk is diagonal row k0 + r (r < 32) or own row t0 + r - 32; rows without data (a short last panel, the tail of the last workgroup) read
    // some valid row instead: a row of garbage only reaches outputs that are dropped.  The rhs row's y (stride ld) is an extra load of the wavefront it
    // belongs to.  Every load is UNCONDITIONAL (a branch around a load costs a full wait per element); the f64 matrix-core instruction blocks the vector
    // pipeline for its 64 cycles, so every VALU instruction saved here is time saved.
    const int sc = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = sc;
    const double* rowp[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int r = wave + 4 * u, t = t0 + r - BA_CH_NB;
        rowp[u] = A + (size_t)(r < BA_CH_NB ? k0 + min(r, nb - 1) : (t < mrows ? r0 + t : 0)) * ld;
    }
    const int t_rhs = mrows - t0;  // own-row index of the rhs row, if this workgroup has it
    const bool has_rhs = t_rhs >= 0 && t_rhs < BA_CL_RB && (t_rhs & 3) == wave;  // (wave uniform: rows of sX are dealt to wavefronts modulo 4)
    double pre[32], yv[2] = {0.0, 0.0};
    auto fetch = [&](int kc) {
        const int c0 = min(sc, k0 - 1 - kc), c1 = min(sc + 64, k0 - 1 - kc);  // (a ragged last chunk re-reads its last column; zeroed on the way to LDS)
#pragma unroll
        for (int u = 0; u < 16; u++) { pre[2 * u] = rowp[u][kc + c0]; pre[2 * u + 1] = rowp[u][kc + c1]; }
        yv[0] = A[(size_t)(kc + c0) * ld + nq]; yv[1] = A[(size_t)(kc + c1) * ld + nq];
    };
    // the products run on the matrix cores (v_mfma_f64_16x16x4_f64: C[r][c] += sum_k X[r][k] X[c][k], both operands straight from the staged rows; the
    // VALU form needed 6 bytes of LDS traffic per multiply-add and was bound by it).  Wavefront w owns output rows 16 w .. 16 w + 15 (w < 2: the diagonal
    // block, w >= 2: this workgroup's rows), tile columns 0 and 1.  Lane: row (lane >> 4) + 4 rg of the tile, column lane & 15.  Two accumulators per
    // tile (even / odd slabs).
    double4v acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = double4v{0.0, 0.0, 0.0, 0.0};
    // the entries the products are subtracted from (independent of the loop: their round trip overlaps it); unconditional loads, masked
    double org[2][4];
#pragma unroll
    for (int tc = 0; tc < 2; tc++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int r = 16 * wave + (lane >> 4) + 4 * rg, c = 16 * tc + (lane & 15), t = t0 + r - BA_CH_NB;
            const int cl = min(c, nb - 1);
            unsigned idx = 0u, m = c < nb ? ~0u : 0u;
            if (r < BA_CH_NB) { idx = (unsigned)((k0 + min(r, nb - 1)) * ld + k0 + cl); m &= (r < nb && c <= r) ? ~0u : 0u; }
            else if (t < mrows) idx = (unsigned)((r0 + t) * ld + k0 + cl);
            else if (t == mrows) idx = (unsigned)((k0 + cl) * ld + nq);
            else m = 0u;
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, A[idx]);
            org[tc][rg] = __builtin_bit_cast(double, bits & (((unsigned long long)m << 32) | m));
        }
    if (k0 > 0) fetch(0);
    for (int kc = 0; kc < k0; kc += KC) {
        __syncthreads();  // the previous chunk has been read
        if (kc + KC <= k0) {
#pragma unroll
            for (int u = 0; u < 16; u++) { sX[wave + 4 * u][sc] = pre[2 * u]; sX[wave + 4 * u][sc + 64] = pre[2 * u + 1]; }
            if (has_rhs) { sX[BA_CH_NB + t_rhs][sc] = yv[0]; sX[BA_CH_NB + t_rhs][sc + 64] = yv[1]; }
        } else {  // ragged last chunk (k0 is a multiple of 32, a chunk has 128 columns)
            const bool on0 = kc + sc < k0, on1 = kc + sc + 64 < k0;
#pragma unroll
            for (int u = 0; u < 16; u++) { sX[wave + 4 * u][sc] = on0 ? pre[2 * u] : 0.0; sX[wave + 4 * u][sc + 64] = on1 ? pre[2 * u + 1] : 0.0; }
            if (has_rhs) { sX[BA_CH_NB + t_rhs][sc] = on0 ? yv[0] : 0.0; sX[BA_CH_NB + t_rhs][sc + 64] = on1 ? yv[1] : 0.0; }
        }
        __syncthreads();
        if (kc + KC < k0) fetch(kc + KC);
        // (wavefront 0's second tile lies above the diagonal and is never used; skipping it with a branch made hipcc shuffle the accumulators through
        // v_accvgpr_mov behind s_nop 15 in every slab)
        // software pipeline: the operands of step s + 1 are read BEFORE the four matrix-core instructions of step s are issued.  A wavefront blocks at
        // every MFMA issue (64 cycles each), so with the reads after them the LDS round trip of every step was exposed: 52 ns per MFMA instead of 27
        double o[6], n[6];
        const auto read_ops = [&](double (&d)[6], int ks) {
            d[0] = sX[16 * wave + (lane & 15)][ks + (lane >> 4)]; d[1] = sX[16 * wave + (lane & 15)][ks + 4 + (lane >> 4)];
            d[2] = sX[lane & 15][ks + (lane >> 4)]; d[3] = sX[16 + (lane & 15)][ks + (lane >> 4)];
            d[4] = sX[lane & 15][ks + 4 + (lane >> 4)]; d[5] = sX[16 + (lane & 15)][ks + 4 + (lane >> 4)];
        };
        read_ops(o, 0);
#pragma unroll
        for (int ks = 0; ks < KC; ks += 8) {
            if (ks + 8 < KC) read_ops(n, ks + 8);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o[0], o[2], acc[0], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(o[1], o[4], acc[2], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o[0], o[3], acc[1], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(o[1], o[5], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; q++) o[q] = n[q];
        }
    }
    __syncthreads();  // sX is dead: its memory becomes sD / sDt / sV / sDi
#pragma unroll
    for (int tc = 0; tc < 2; tc++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int r = 16 * wave + (lane >> 4) + 4 * rg, c = 16 * tc + (lane & 15);
            const double v = org[tc][rg] - (acc[tc][rg] + acc[tc + 2][rg]);
            if (r < BA_CH_NB) sD[r][c] = (r < nb && c <= r) ? v : (r == c ? 1.0 : 0.0);  // (a short last panel is padded with the identity)
            else sV[r - BA_CH_NB][c] = c < nb ? v : 0.0;
        }
    __syncthreads();
    if (tid < 64) {
        const int i = tid & (BA_CH_NB - 1);  // (lanes 32..63 duplicate lanes 0..31; only the lower 32 are read through v_readlane)
        double a[BA_CH_NB];
#pragma unroll
        for (int j = 0; j < BA_CH_NB; j++) a[j] = sD[i][j];
        ba_chol32_rows(a);
        if (tid < BA_CH_NB) {
#pragma unroll
            for (int j = 0; j < BA_CH_NB; j++) { sD[i][j] = a[j]; sDt[j][i] = a[j]; }
            double d = 0.0;
#pragma unroll
            for (int j = 0; j < BA_CH_NB; j++) d = j == i ? a[j] : d;
            sDi[i] = 1.0 / d;
        }
    }
    __syncthreads();
    // The factor of the diagonal block goes to a block array of its own (J.Spart is free once k_ba_reduce has run), NOT back into A: every workgroup of
    // this launch reads the ORIGINAL block from A, and a workgroup that starts late -- another process's kernels on the device are enough -- would find
    // workgroup 0's factor there instead (a race the point-sharded two-rank test hit in ~30 % of its runs).  k_ba_chol_back reads the blocks from here.
    if (blockIdx.x == 0)
        for (int e = tid; e < BA_CH_NB * BA_CH_NB; e += 256) {
            const int i = e / BA_CH_NB, j = e - i * BA_CH_NB;
            J.Spart[(size_t)k0 * BA_CH_NB + e] = (i < nb && j <= i) ? sD[i][j] : (i == j ? 1.0 : 0.0);
        }
    // own rows: v <- v L_D^-T, one thread per row, right-looking (see k_ba_chol_panel)
    if (tid < cnt) {
        double v[BA_CH_NB];
#pragma unroll
        for (int j = 0; j < BA_CH_NB; j++) v[j] = sV[tid][j];
#pragma unroll
        for (int k = 0; k < BA_CH_NB; k++) {
            v[k] = v[k] * sDi[k];
#pragma unroll
            for (int j = k + 1; j < BA_CH_NB; j++) v[j] = __builtin_fma(-v[k], sDt[k][j], v[j]);
        }
#pragma unroll
        for (int j = 0; j < BA_CH_NB; j++) sV[tid][j] = v[j];
    }
    __syncthreads();
    for (int e = tid; e < cnt * BA_CH_NB; e += 256) {
        const int rr = e / BA_CH_NB, j = e - rr * BA_CH_NB, t = t0 + rr;
        if (j < nb) {
            if (t == mrows) A[(size_t)(k0 + j) * ld + nq] = sV[rr][j];
            else A[(size_t)(r0 + t) * ld + k0 + j] = sV[rr][j];
        }
    }
}

// L^T dc = y, from the last panel to the first (one workgroup of 1024 threads: the product over the rows below a panel is split 32 ways -- with 8 slices of
// 256 threads a thread walked up to 92 rows of dependent loads, 160 us per solve at 768 unknowns)
#define BA_CB_THREADS 1024
__global__ __launch_bounds__(BA_CB_THREADS) void k_ba_chol_back(BaJob J, int blocks_apart)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nq = J.nq, ld = nq + 1, tid = threadIdx.x;
    const double* A = J.Sfull;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sx = reinterpret_cast<double*>(smem);  // [nq] solution so far
    constexpr int NP = BA_CB_THREADS / BA_CH_NB;   // row slices
    __shared__ double sT[NP][BA_CH_NB + 1];
    __shared__ double sD[BA_CH_NB][BA_CH_NB + 1];
    __shared__ double st[BA_CH_NB];
    const int npan = (nq + BA_CH_NB - 1) / BA_CH_NB;
    for (int p = npan - 1; p >= 0; p--) {
        const int k0 = p * BA_CH_NB, nb = min(BA_CH_NB, nq - k0), r0 = k0 + nb;
        // t[j] = y[k0 + j] - sum_{i >= r0} L[i][k0 + j] x[i]: NP row slices x 32 columns, four independent chains per thread
        const int j = tid & (BA_CH_NB - 1), part = tid >> 5;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (j < nb) {
            int i = r0 + part;
            for (; i + 3 * NP < nq; i += 4 * NP) {
                const double l0 = A[(size_t)i * ld + k0 + j], l1 = A[(size_t)(i + NP) * ld + k0 + j], l2 = A[(size_t)(i + 2 * NP) * ld + k0 + j],
                             l3 = A[(size_t)(i + 3 * NP) * ld + k0 + j];
                a0 = __builtin_fma(l0, sx[i], a0); a1 = __builtin_fma(l1, sx[i + NP], a1);
                a2 = __builtin_fma(l2, sx[i + 2 * NP], a2); a3 = __builtin_fma(l3, sx[i + 3 * NP], a3);
            }
            for (; i < nq; i += NP) a0 = __builtin_fma(A[(size_t)i * ld + k0 + j], sx[i], a0);
        }
        sT[part][j] = (a0 + a1) + (a2 + a3);
        for (int e = tid; e < BA_CH_NB * BA_CH_NB; e += BA_CB_THREADS) {
            const int i = e / BA_CH_NB, jj = e - i * BA_CH_NB;
            // diagonal factor blocks: in their own array after k_ba_chol_left (see there), in place after the right-looking kernels
            sD[i][jj] = blocks_apart ? J.Spart[(size_t)k0 * BA_CH_NB + e] : ((i < nb && jj <= i) ? A[(size_t)(k0 + i) * ld + k0 + jj] : (i == jj ? 1.0 : 0.0));
        }
        __syncthreads();
        if (tid < BA_CH_NB) {
            double t = tid < nb ? A[(size_t)(k0 + tid) * ld + nq] : 0.0;
            for (int q = 0; q < NP; q++) t -= sT[q][tid];
            st[tid] = t;
        }
        __syncthreads();
        // L_D^T x = t: x[j] from the last to the first; after x[j], every earlier entry drops its term.  One wavefront, lane i holds column i of L_D and t_i;
        // x_j travels by v_readlane (straight-line code; the LDS version with two wave barriers per step cost 4 us per panel)
        if (tid < 64) {
            const int i = tid & (BA_CH_NB - 1);
            double c[BA_CH_NB];
#pragma unroll
            for (int jj = 0; jj < BA_CH_NB; jj++) c[jj] = sD[jj][i];  // L_D[jj][i]: zero above the diagonal (jj < i)
            double t = st[i];
            const double rinv = 1.0 / sD[i][i];
#pragma unroll
            for (int jj = BA_CH_NB - 1; jj >= 0; jj--) {
                const double xj = ba_readlane_f64(t, jj) * ba_readlane_f64(rinv, jj);
                t = i == jj ? xj : __builtin_fma(-c[jj], xj, t);
            }
            if (tid < BA_CH_NB) st[i] = t;
        }
        __syncthreads();
        if (tid < nb) sx[k0 + tid] = st[tid];
        __syncthreads();
    }
    for (int q = tid; q < nq; q += BA_CB_THREADS) J.dc[q] = sx[q];
}

// tail of the update kernels: block 0 moves the cameras (and builds the rotation / offset tables of the next iteration from the LDS copy of the new
// parameters -- saves a launch per iteration); every block adds its sum of squared steps, the last one finishes the iteration record
template <int NT>
__device__ __forceinline__ void ba_update_tail(BaJob& J, int it, double ss, double* s_par, double* sh)
{
    const int nt = J.nt, nc = J.nc, nq = J.nq, tid = threadIdx.x;
    if (blockIdx.x == 0)
        for (int q = tid; q < nq; q += NT) {
            const int c = q / 6, k = q - 6 * c;
            const double dl = J.dc[q] * 0.9;
            // state layout: [points | camera positions | camera rpy] (NLS.py:203); model 1: [points | rpy, el, az, ranges] in reduced order
            const size_t idx = J.model == 1 ? (size_t)3 * nt + q
                                            : (k < 3 ? (size_t)3 * nt + 3 * c + k : (size_t)3 * nt + 3 * nc + 3 * c + (k - 3));
            const double nv = J.x[idx] + dl;
            J.x[idx] = nv;
            s_par[idx - (size_t)3 * nt] = nv;
            if (J.count_cams) ss += dl * dl;  // sharded runs: the (replicated) camera update is counted by rank 0 only
        }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (J.model == 0) for (int cv = tid; cv < 4 * (nc + 1); cv += NT) ba_cam_table_variant(J, cv >> 2, cv & 3, s_par);
        else for (int c = tid; c <= nc; c += NT) ba_cam_tables(J, c, s_par);
    }
    ss = vh_wave_sum_f64(ss);
    if ((tid & 63) == 0) sh[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < NT / 64; k++) s += sh[k];
        atomicAdd(J.acc + 1, s);
        __threadfence();
        const unsigned prev = atomicAdd(J.ticket, 1u);
        if (prev == gridDim.x - 1 && J.defer_finalize) *J.ticket = 0u;
        if (prev == gridDim.x - 1 && !J.defer_finalize) {  // last block: finish the iteration record
            const double nz = J.nz_total, nx = J.nx_total;
            const double sumr = atomicAdd(J.acc, 0.0), sumd = atomicAdd(J.acc + 1, 0.0);
            const double f = sqrt(sumr / nz), xr = sqrt(sumd / nx);
            J.trace[2 * it] = f;
            J.trace[2 * it + 1] = xr;
            J.info[0] = it + 1;
            if (xr < 1e-7) { J.info[1] = 1; *J.done = 1; }
            J.acc[0] = 0.0; J.acc[1] = 0.0;
            *J.ticket = 0u;
            __threadfence();
        }
    }
}

// back-substitution dp = tp - Y dc, update x += 0.9 delta, rms(delta) and the stop flag (NLS.py:235-240): VALU Schur path (zmode = 0; k_ba_update_z otherwise)
__global__ __launch_bounds__(BA_THREADS) void k_ba_update(BaJob J, int it)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = J.nq, tid = threadIdx.x;
    __shared__ double sh[BA_THREADS / 64];
    __shared__ double s_par[6 * BA_MAX_NC];  // new camera-side parameters (block 0)
    double ss = 0.0;
    const int lane = tid & 63, wave = tid >> 6;
    {
        // VALU path: one wavefront per point: the lanes split the 6nc columns of the point's 3 x 6nc block of Y (consecutive lanes read
        // consecutive 24-byte column triples -> coalesced; a thread-per-point walk of the 2.7 KB rows ran at 0.3 TB/s)
        for (int i = blockIdx.x * (BA_THREADS / 64) + wave; i < nt; i += gridDim.x * (BA_THREADS / 64)) {
            const double* Y = J.Y + (size_t)i * nq * 3;
            double d0 = 0.0, d1 = 0.0, d2 = 0.0;
            for (int q = lane; q < nq; q += 64) {
                const double dq = J.dc[q];
                d0 += Y[3 * q] * dq; d1 += Y[3 * q + 1] * dq; d2 += Y[3 * q + 2] * dq;
            }
            d0 = vh_wave_sum_f64(d0); d1 = vh_wave_sum_f64(d1); d2 = vh_wave_sum_f64(d2);
            if (lane == 0) {
                const double d[3] = {J.tp[3 * (size_t)i] - d0, J.tp[3 * (size_t)i + 1] - d1, J.tp[3 * (size_t)i + 2] - d2};
                for (int k = 0; k < 3; k++) {
                    const double dl = d[k] * 0.9;
                    J.x[3 * (size_t)i + k] += dl;
                    ss += dl * dl;
                }
            }
        }
    }
    ba_update_tail<BA_THREADS>(J, it, ss, s_par, sh);
}

// matrix-core paths (zmode = 1): dp_i = tp_i - (U_i+I)^-1 W_i dc with W_i dc = sum_c Jp_c^T (Jc_c dc_c) recomputed from the compact Jacobians (the same bytes a
// stored Y / Z would cost to read).  32 lanes share a point, ONE camera per lane and pass (cameras c = 1 + sub, 33 + sub, ...): the 18 doubles of a
// measurement arrive as nine 16-byte loads issued together -- one memory round trip per point instead of the five dependent ones of the 4-lanes-per-point
// loop (round 5: 20 -> ~9 us for one C5 window, whose Jacobians were written by other XCDs in the previous launch and come from the infinity cache / HBM)
// Up to 20 000 points per launch (latency regime); more take k_ba_update_z4.
#define BA_UZ_LPP 32  // lanes per point
template <int BA_UZ_THREADS>
__global__ __launch_bounds__(BA_UZ_THREADS) void k_ba_update_z(BaJob J, int it)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = J.nq, tid = threadIdx.x;
    __shared__ double sh[BA_UZ_THREADS / 64];
    __shared__ double s_par[6 * BA_MAX_NC];  // dc, then the new camera-side parameters (block 0)
    double ss = 0.0;
    for (int q = tid; q < nq; q += BA_UZ_THREADS) s_par[q] = J.dc[q];
    __syncthreads();
    typedef double double2v __attribute__((ext_vector_type(2)));
    constexpr int PPB = BA_UZ_THREADS / BA_UZ_LPP;  // points per block and pass
    const int sub = tid & (BA_UZ_LPP - 1), pl = tid / BA_UZ_LPP;
    // block 0 owns the cameras and no point: its chain (dc -> new parameters -> 6 sin / cos per rotation table) runs beside the point blocks, not after
    // block 0's share of them
    const int npb = (int)gridDim.x - 1, pb = (int)blockIdx.x - 1;
    const int npass = pb < 0 ? 0 : (nt + PPB * npb - 1) / (PPB * npb);  // whole blocks run the loop together (the shuffles below need their partners)
    for (int ps = 0; ps < npass; ps++) {
        const int ip = (ps * npb + pb) * PPB + pl, i = min(ip, nt - 1);
        double e0 = 0.0, e1 = 0.0, e2 = 0.0;
        for (int c = 1 + sub; c <= nc; c += BA_UZ_LPP) {
            const size_t m = (size_t)i * (nc + 1) + c;
            const double2v* Jc = reinterpret_cast<const double2v*>(J.Jc + 12 * m);  // 96 m bytes: 16-byte aligned
            const double2v* Jp = reinterpret_cast<const double2v*>(J.Jp + 6 * m);   // 48 m bytes
            const double2v a0 = Jc[0], a1 = Jc[1], a2 = Jc[2], b0 = Jc[3], b1 = Jc[4], b2 = Jc[5], p0 = Jp[0], p1 = Jp[1], p2 = Jp[2];
            const double* dq = s_par + 6 * (c - 1);
            const double su = a0.x * dq[0] + a0.y * dq[1] + a1.x * dq[2] + a1.y * dq[3] + a2.x * dq[4] + a2.y * dq[5];
            const double sv = b0.x * dq[0] + b0.y * dq[1] + b1.x * dq[2] + b1.y * dq[3] + b2.x * dq[4] + b2.y * dq[5];
            // Jp = [du/dX du/dY du/dZ | dv/dX dv/dY dv/dZ]
            e0 += p0.x * su + p1.y * sv; e1 += p0.y * su + p2.x * sv; e2 += p1.x * su + p2.y * sv;
        }
#pragma unroll
        for (int o = 1; o < BA_UZ_LPP; o <<= 1) { e0 += __shfl_xor(e0, o, 64); e1 += __shfl_xor(e1, o, 64); e2 += __shfl_xor(e2, o, 64); }
        if (sub == 0 && ip < nt) {
            const double* Lp = J.Lc + 6 * (size_t)i;  // (U+I)^-1 = L L^T, L = l00 l10 l11 l20 l21 l22
            const double z0 = Lp[0] * e0 + Lp[1] * e1 + Lp[3] * e2, z1 = Lp[2] * e1 + Lp[4] * e2, z2 = Lp[5] * e2;  // L^T e
            const double d0 = Lp[0] * z0, d1 = Lp[1] * z0 + Lp[2] * z1, d2 = Lp[3] * z0 + Lp[4] * z1 + Lp[5] * z2;  // L (L^T e)
            const double d[3] = {J.tp[3 * (size_t)i] - d0, J.tp[3 * (size_t)i + 1] - d1, J.tp[3 * (size_t)i + 2] - d2};
            for (int k = 0; k < 3; k++) {
                const double dl = d[k] * 0.9;
                J.x[3 * (size_t)i + k] += dl;
                ss += dl * dl;
            }
        }
    }
    __syncthreads();  // s_par is reused for the camera parameters
    ba_update_tail<BA_UZ_THREADS>(J, it, ss, s_par, sh);
}

// zmode = 1, more than 20 000 points per launch (bandwidth regime): 4 lanes share a point (cameras c = 1 + sub, 5 + sub, ...), every lane busy.  (k_ba_update_z with its
// 19 of 32 lanes at C5 ran 282 us against 241 at 64 windows.)
__global__ __launch_bounds__(BA_THREADS) void k_ba_update_z4(BaJob J, int it)
{
    ba_select_window(J, blockIdx.y);
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = J.nq, tid = threadIdx.x;
    __shared__ double sh[BA_THREADS / 64];
    __shared__ double s_par[6 * BA_MAX_NC];
    double ss = 0.0;
    for (int q = tid; q < nq; q += BA_THREADS) s_par[q] = J.dc[q];
    __syncthreads();
    const int total = 4 * ((nt + 15) / 16) * 16;  // whole waves run the loop together (the shuffles below need their 4 partners)
    for (int p4 = blockIdx.x * BA_THREADS + tid; p4 < total; p4 += gridDim.x * BA_THREADS) {
        const int i = min(p4 >> 2, nt - 1), sub = p4 & 3;
        double e0 = 0.0, e1 = 0.0, e2 = 0.0;
        for (int c = 1 + sub; c <= nc; c += 4) {
            const size_t m = (size_t)i * (nc + 1) + c;
            const double* Jc = J.Jc + 12 * m;
            const double* Jp = J.Jp + 6 * m;
            const double* dq = s_par + 6 * (c - 1);
            double su = 0.0, sv = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) { su += Jc[k] * dq[k]; sv += Jc[6 + k] * dq[k]; }
            e0 += Jp[0] * su + Jp[3] * sv; e1 += Jp[1] * su + Jp[4] * sv; e2 += Jp[2] * su + Jp[5] * sv;
        }
        e0 += __shfl_xor(e0, 1, 64); e1 += __shfl_xor(e1, 1, 64); e2 += __shfl_xor(e2, 1, 64);
        e0 += __shfl_xor(e0, 2, 64); e1 += __shfl_xor(e1, 2, 64); e2 += __shfl_xor(e2, 2, 64);
        if (sub == 0 && (p4 >> 2) < nt) {
            const double* Lp = J.Lc + 6 * (size_t)i;  // (U+I)^-1 = L L^T, L = l00 l10 l11 l20 l21 l22
            const double z0 = Lp[0] * e0 + Lp[1] * e1 + Lp[3] * e2, z1 = Lp[2] * e1 + Lp[4] * e2, z2 = Lp[5] * e2;  // L^T e
            const double d0 = Lp[0] * z0, d1 = Lp[1] * z0 + Lp[2] * z1, d2 = Lp[3] * z0 + Lp[4] * z1 + Lp[5] * z2;  // L (L^T e)
            const double d[3] = {J.tp[3 * (size_t)i] - d0, J.tp[3 * (size_t)i + 1] - d1, J.tp[3 * (size_t)i + 2] - d2};
            for (int k = 0; k < 3; k++) {
                const double dl = d[k] * 0.9;
                J.x[3 * (size_t)i + k] += dl;
                ss += dl * dl;
            }
        }
    }
    __syncthreads();  // s_par is reused for the camera parameters
    ba_update_tail<BA_THREADS>(J, it, ss, s_par, sh);
}

// iteration record from the (all-reduced) sums of a sharded run
__global__ void k_ba_finalize(BaJob J, int it)
{
    ba_select_window(J, blockIdx.y);
    if (threadIdx.x != 0 || *J.done) return;
    const double f = sqrt(J.acc[0] / J.nz_total), xr = sqrt(J.acc[1] / J.nx_total);
    J.trace[2 * it] = f;
    J.trace[2 * it + 1] = xr;
    J.info[0] = it + 1;
    if (xr < 1e-7) { J.info[1] = 1; *J.done = 1; }
    J.acc[0] = 0.0; J.acc[1] = 0.0;
}

// start of a solve: accumulators, flag block and the info record of every window
__global__ void k_ba_init(BaJob J, double* flags0)
{
    ba_select_window(J, blockIdx.y);
    double* flags = reinterpret_cast<double*>(reinterpret_cast<char*>(flags0) + (size_t)blockIdx.y * (J.nwin > 1 ? J.ws_stride : 0));
    const int t = threadIdx.x;
    if (t < 4) J.acc[t] = 0.0;
    if (t < 32) flags[t] = 0.0;
    if (t < 2) J.info[t] = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// replayable launch sequences (hipGraph) of whole solves, owned by the vh_ctx that issues them
struct BaGraphKey {
    BaJob J;
    double* flags0;
    int max_iter, nparts, use_mfma, pad_;
};
struct BaGraphEntry {
    BaGraphKey key;
    hipGraphExec_t exec;
    int seen;
    unsigned long long stamp;
};
struct BaGraphCache {
    static constexpr int CAP = 8;
    BaGraphEntry e[CAP];
    int n = 0;
    unsigned long long clock = 0;
    bool disabled = false;
    hipStream_t capture_stream = nullptr;
    BaGraphEntry* find(const BaGraphKey& k, hipStream_t s)
    {
        for (int i = 0; i < n; i++)
            if (memcmp(&e[i].key, &k, sizeof(k)) == 0) { e[i].stamp = ++clock; return &e[i]; }
        BaGraphEntry* slot = nullptr;
        if (n < CAP) slot = &e[n++];
        else {
            slot = &e[0];
            for (int i = 1; i < CAP; i++)
                if (e[i].stamp < slot->stamp) slot = &e[i];
            if (slot->exec) {
                (void)hipStreamSynchronize(s);  // the evicted sequence may still be running
                (void)hipGraphExecDestroy(slot->exec);
            }
        }
        memcpy(&slot->key, &k, sizeof(k)); slot->exec = nullptr; slot->seen = 0; slot->stamp = ++clock;
        return slot;
    }
};
static BaGraphCache* ba_graph_cache(void** where)
{
    static const bool off = [] { const char* v = getenv("VH_BA_GRAPH"); return v && v[0] == '0'; }();
    if (!where || off) return nullptr;
    if (!*where) {
        BaGraphCache* c = new (std::nothrow) BaGraphCache();
        if (!c) return nullptr;
        if (hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); delete c; return nullptr; }
        *where = c;
    }
    return static_cast<BaGraphCache*>(*where);
}
void vh_ba_graph_cache_free(void* cache)
{
    BaGraphCache* c = static_cast<BaGraphCache*>(cache);
    if (!c) return;
    for (int i = 0; i < c->n; i++)
        if (c->e[i].exec) (void)hipGraphExecDestroy(c->e[i].exec);
    if (c->capture_stream) (void)hipStreamDestroy(c->capture_stream);
    delete c;
}

size_t vh_ba_workspace_bytes(int nt, int nc, int nparts)
{
    const size_t nf = nc + 1, nq = 6 * (size_t)nc, m = (size_t)nt * nf;
    size_t b = 0;
    auto add = [&](size_t n) { b += (n * sizeof(double) + 255) / 256 * 256; };
    const bool big = nq > 252;  // 43+ cameras: per-workgroup partials of k_ba_zbuild (see there)
    add(36 * nf); add(2 * m); add(6 * m); add(12 * m); add(3 * (size_t)nt); add(6 * (size_t)nt); add(3 * nq * nt); add(nparts * nq * nq);
    add((big ? std::max((size_t)nparts, (size_t)BA_ZB_MAX) : (size_t)nparts) * nq); add(big ? (size_t)BA_ZB_MAX * 36 * nc : 0);
    add(nq * (nq + 1) + 4); add(nq); add(32);
    return b + 1024;
}

static void ba_layout(const BaProblem& P, BaJob& J, double*& flags)
{
    const int nt = P.nt, nc = P.nc, nq = 6 * nc, nparts = P.nparts;
    J.nt = nt; J.nc = nc;
    J.model = P.model; J.nq = P.model == 1 ? nc + 5 : 6 * nc;  // buffers are sized for 6 nc >= nc + 5
    for (int k = 0; k < 9; k++) J.K[k] = P.K[k];
    J.z = P.z; J.x = P.x; J.trace = P.trace; J.info = P.info;
    J.add_identity = P.add_identity; J.count_cams = P.count_cams; J.defer_finalize = P.defer_finalize;
    J.nx_total = P.nx_total; J.nz_total = P.nz_total;
    J.nwin = P.nwin < 1 ? 1 : P.nwin;
    J.ws_stride = P.ws_stride; J.z_stride = P.z_stride; J.x_stride = P.x_stride; J.trace_stride = P.trace_stride; J.info_stride = P.info_stride;
    char* w = reinterpret_cast<char*>(P.workspace);
    auto take = [&](size_t n) { double* p = reinterpret_cast<double*>(w); w += (n * sizeof(double) + 255) / 256 * 256; return p; };
    const size_t nf = nc + 1, m = (size_t)nt * nf;
    J.camR = take(36 * nf); J.r = take(2 * m); J.Jp = take(6 * m); J.Jc = take(12 * m); J.tp = take(3 * (size_t)nt); J.Lc = take(6 * (size_t)nt); J.Y = take(3 * (size_t)nq * nt);
    const bool big = nq > 252;
    J.Spart = take((size_t)nparts * nq * nq); J.Rpart = take((big ? std::max((size_t)nparts, (size_t)BA_ZB_MAX) : (size_t)nparts) * nq);
    J.Dpart = take(big ? (size_t)BA_ZB_MAX * 36 * nc : 0);
    J.Sfull = take((size_t)nq * (nq + 1) + 4);     // augmented system followed by the 4 accumulators: ONE all-reduce span
    J.acc = J.Sfull + (size_t)nq * (nq + 1);
    J.dc = take(nq);
    flags = take(32);
    J.done = reinterpret_cast<int*>(flags);
    J.ticket = reinterpret_cast<unsigned*>(flags) + 4;
    J.rslot = flags + 8;  // doubles 8..23 of the 32-double flag block
}

// byte offset / length (in doubles) of the all-reduce span [Sfull | acc] inside the workspace
void vh_ba_exchange_span(const BaProblem& P, size_t* offset_bytes, size_t* n_doubles)
{
    BaJob J;
    double* flags;
    ba_layout(P, J, flags);
    *offset_bytes = (size_t)(reinterpret_cast<char*>(J.Sfull) - reinterpret_cast<char*>(P.workspace));
    *n_doubles = (size_t)(6 * P.nc) * (6 * P.nc + 1) + 4;
}

// phase -1: the whole solve on one rank.  Sharded runs (velocity_amd/dist.py): 0 = init, 1 = local normal equations ->
// [Sfull | acc] (then all-reduced by the caller), 2 = solve + update (then acc all-reduced), 3 = iteration record.
int vh_ba_run(const BaProblem& P, hipStream_t s)
{
    const int nt = P.nt, nc = P.nc, nq = P.model == 1 ? nc + 5 : 6 * nc;
    if (nc > BA_MAX_NC) return -3;
    const int nparts = P.nparts;
    BaJob J;
    memset(&J, 0, sizeof(J));  // padding included: whole solves are recognised by the bytes of their descriptor (graph replay below)
    double* flags;
    ba_layout(P, J, flags);
    const long long nent = (long long)nq * nq;
    const int npass = (int)((nent + (long long)BA_THREADS * BA_EPT - 1) / ((long long)BA_THREADS * BA_EPT));
    const size_t lds = sizeof(double) * (size_t)(6 * nq + 12 * (nc + 1) + 16);
    const int npad = nq <= BA_NPAD ? BA_NPAD : 2 * BA_NPAD;  // matrix-core Schur kernel: 128-wide (<= 21 cameras) or 256-wide in two passes (<= 42)
    const size_t lds_mfma = sizeof(double) * (size_t)(24 * npad + 2 * 4 * (20 * nc + 10) + 4 * npad);
    // the VALU test hook (vh_debug_ba_force_valu) is honoured where the VALU Schur kernel exists (<= BA_MAX_NC_VALU cameras) and ignored above: a flag
    // left set by another test or thread must not turn a valid call into a failure
    const bool force_valu = P.force_valu && nq <= 6 * BA_MAX_NC_VALU;
    const bool use_mfma = nq <= 252 && !force_valu && P.model == 0;  // model 1 has nc + 5 unknowns: nothing for the matrix cores to do
    // 43..128 cameras: Z materialised + K-split SYRK on the matrix cores (k_ba_zbuild / k_ba_syrk_mfma).  nm macro tiles of 128 per dimension; the
    // first `nsplit` partial systems are used: about one resident round of workgroups (2 per CU), every split at least one LDS stage of points
    const bool use_syrk = nq > 252 && !force_valu && P.model == 0;
    if (!use_syrk && !use_mfma && nq > 6 * BA_MAX_NC_VALU) return -3;  // the VALU Schur kernel keeps 6 nc / 256 right-hand-side entries per thread and 3 x 6 nc doubles of LDS
    const int nm = (nq + 127) / 128, npairs = nm * (nm + 1) / 2;
    const int nsplit = use_syrk ? std::max(1, std::min(std::min(nparts, std::max(1, 512 / (npairs * (int)std::min(J.nwin < 1 ? 1 : J.nwin, 512)))), (nt + 10) / 11)) : nparts;
    const int nzb = use_syrk ? std::max(1, std::min(BA_ZB_MAX, (nt + 2 * BA_ZB_PL - 1) / (2 * BA_ZB_PL))) : 0;  // workgroups of k_ba_zbuild: at least two rounds of points each
    if ((use_mfma && lds_mfma > 64 * 1024) || nq > BA_GJ_MAXQ) {
        // the attribute is per function AND per device: one bit per device ordinal, set with an atomic OR (two host threads, or a process that drives
        // a second GPU, each set it for their device; setting it twice is harmless)
        static std::atomic<unsigned long long> attr_devs{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_schur_mfma<256, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_schur_mfma<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_chol_left), hipFuncAttributeMaxDynamicSharedMemorySize, BA_CL_LDS);
            if (e != hipSuccess) return (int)e;
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
    }
    J.zmode = (use_mfma || use_syrk) ? 1 : 0;
    { const char* e = getenv("VH_BA_DBG"); J.dbg = e ? atoi(e) : 0; }
    // one wavefront per point, grid-strided: every block ends with an atomic on one address, so keep the block count low
    const int upd_cap = J.nwin > 1 ? std::max(16, 1024 / J.nwin) : 256;
    const bool uz_latency = (long long)J.nwin * nt <= 20000;  // few points in flight: the 32-lanes-per-point kernel (one memory round trip); else every lane busy
    const int uz_ppb = uz_latency ? 1024 / BA_UZ_LPP : BA_THREADS / 4;  // points per block and pass of k_ba_update_z / k_ba_update_z4
    const int upd_blocks = std::min((use_mfma || use_syrk) ? (nt + uz_ppb - 1) / uz_ppb : (nt + BA_THREADS / 64 - 1) / (BA_THREADS / 64), upd_cap);
    const unsigned nw = (unsigned)J.nwin;
    auto init = [&]() -> int {
        hipLaunchKernelGGL(k_ba_init, dim3(1, nw), dim3(64), 0, s, J, flags);
        return (int)hipGetLastError();
    };
    vh_ctx* pc = P.ctx;  // per-kernel event timing (bench.py's ba.roofline); profiled solves are launched plainly, never replayed
    const bool profiling = pc && pc->prof_on;
    auto normal_equations = [&](int it) {
        if (it == 0) hipLaunchKernelGGL(k_ba_cams, dim3((nc + 1 + 63) / 64, nw), dim3(64), 0, s, J);
        const int ppb = BA_THREADS / (nc + 1);  // k_ba_jac: whole points per block
        if (use_mfma) {
            int rec = vh_prof_start(pc, s);
            hipLaunchKernelGGL(k_ba_jac<true>, dim3((nt + ppb - 1) / ppb, nw), dim3(BA_THREADS), 0, s, J);
            vh_prof_stop(pc, rec, VH_PROF_BA_JAC, s);
            rec = vh_prof_start(pc, s);
            if (npad == BA_NPAD) hipLaunchKernelGGL((k_ba_schur_mfma<128, 0>), dim3(nparts, nw), dim3(BA_SCHUR_THREADS), lds_mfma, s, J);
            else {
                hipLaunchKernelGGL((k_ba_schur_mfma<256, 0>), dim3(nparts, nw), dim3(BA_SCHUR_THREADS), lds_mfma, s, J);
                hipLaunchKernelGGL((k_ba_schur_mfma<256, 1>), dim3(nparts, nw), dim3(BA_SCHUR_THREADS), lds_mfma, s, J);
            }
            vh_prof_stop(pc, rec, VH_PROF_BA_SCHUR, s);
        } else if (use_syrk) {
            int rec = vh_prof_start(pc, s);
            hipLaunchKernelGGL(k_ba_jac<true>, dim3((nt + ppb - 1) / ppb, nw), dim3(BA_THREADS), 0, s, J);
            vh_prof_stop(pc, rec, VH_PROF_BA_JAC, s);
            rec = vh_prof_start(pc, s);
            hipLaunchKernelGGL(k_ba_zbuild, dim3(nzb, (nq + 255) / 256, nw), dim3(256, BA_ZB_PL), 0, s, J);
            hipLaunchKernelGGL(k_ba_syrk_mfma, dim3(npairs, nsplit, nw), dim3(256), 0, s, J, nm);
            vh_prof_stop(pc, rec, VH_PROF_BA_SCHUR, s);
        } else {
            int rec = vh_prof_start(pc, s);
            hipLaunchKernelGGL(k_ba_jac<false>, dim3((nt + ppb - 1) / ppb, nw), dim3(BA_THREADS), 0, s, J);
            vh_prof_stop(pc, rec, VH_PROF_BA_JAC, s);
            rec = vh_prof_start(pc, s);
            for (int pass = 0; pass < npass; pass++)  // later passes overwrite Spart entries of their own range only
                hipLaunchKernelGGL(k_ba_points, dim3(nparts, nw), dim3(BA_THREADS), lds, s, J, pass);
            vh_prof_stop(pc, rec, VH_PROF_BA_SCHUR, s);
        }
        const int rec = vh_prof_start(pc, s);
        if (nsplit >= 128) hipLaunchKernelGGL(k_ba_reduce<16>, dim3((unsigned)((nent + nq + 63) / 64), nw), dim3(1024), 0, s, J, nsplit, use_syrk ? nzb : nsplit, use_syrk ? nzb : 0);
        else hipLaunchKernelGGL(k_ba_reduce<4>, dim3((unsigned)((nent + nq + 63) / 64), nw), dim3(256), 0, s, J, nsplit, use_syrk ? nzb : nsplit, use_syrk ? nzb : 0);
        vh_prof_stop(pc, rec, VH_PROF_BA_REDUCE, s);
    };
    auto solve_update = [&](int it) {
        int rec = vh_prof_start(pc, s);
        // up to 124 unknowns (20 cameras): block Gauss-Jordan on the matrix cores; 125..127 (21 cameras: 126): the register-resident VALU Gauss-Jordan
        if (nq <= BA_GJ_MAXQ && !(J.dbg & 64)) hipLaunchKernelGGL(k_ba_solve_mfma, dim3(1, nw), dim3(64 * BA_GJ_WAVES), 0, s, J);
        else if (nq + 1 <= 128) hipLaunchKernelGGL((k_ba_solve<8, 8, 16>), dim3(1, nw), dim3(256), 0, s, J, nparts);
        else {  // 128+ unknowns (22+ cameras): left-looking blocked Cholesky, ONE launch per 32-column panel, + the back-substitution
            for (int k0 = 0; k0 < nq && !(J.dbg & 256); k0 += BA_CH_NB) {  // left-looking: one launch per panel, rows spread over workgroups
                const int rows = nq - std::min(nq, k0 + BA_CH_NB) + 1;
                hipLaunchKernelGGL(k_ba_chol_left, dim3((rows + BA_CL_RB - 1) / BA_CL_RB, nw), dim3(256), BA_CL_LDS, s, J, k0);
            }
            for (int k0 = 0; k0 < nq && (J.dbg & 256); k0 += BA_CH_NB) {  // right-looking kernels of round 4 (VH_BA_DBG=256: second implementation for the tests)
                hipLaunchKernelGGL(k_ba_chol_panel, dim3(1, nw), dim3(256), 0, s, J, k0);
                const int r0 = std::min(nq, k0 + BA_CH_NB), ntile = (nq - r0 + BA_CH_NB - 1) / BA_CH_NB;
                if (ntile > 0) hipLaunchKernelGGL(k_ba_chol_update, dim3(ntile + 1, ntile, nw), dim3(256), 0, s, J, k0);
            }
            hipLaunchKernelGGL(k_ba_chol_back, dim3(1, nw), dim3(BA_CB_THREADS), sizeof(double) * (size_t)nq, s, J, (J.dbg & 256) ? 0 : 1);
        }
        vh_prof_stop(pc, rec, VH_PROF_BA_SOLVE, s);
        rec = vh_prof_start(pc, s);
        if (J.zmode && uz_latency) hipLaunchKernelGGL(k_ba_update_z<1024>, dim3(upd_blocks + 1, nw), dim3(1024), 0, s, J, it);  // (+ 1: the camera block)
        else if (J.zmode) hipLaunchKernelGGL(k_ba_update_z4, dim3(upd_blocks, nw), dim3(BA_THREADS), 0, s, J, it);
        else hipLaunchKernelGGL(k_ba_update, dim3(upd_blocks, nw), dim3(BA_THREADS), 0, s, J, it);
        vh_prof_stop(pc, rec, VH_PROF_BA_UPDATE, s);
    };
    switch (P.phase) {
    case -1: {
        // whole solve: 2 + 5 max_iter dependent launches.  A sequence seen before (same job descriptor -- pointers, sizes, intrinsics -- and launch
        // shape) is replayed as ONE hipGraph launch: the device then runs its kernels back to back whatever the host's launch rate is (a busy or
        // throttled host otherwise shows up as idle gaps between the 10-60 us kernels of a single window).  The second sighting builds the graph.
        BaGraphKey key;
        memset(&key, 0, sizeof(key));
        memcpy(&key.J, &J, sizeof(J)); key.flags0 = flags; key.max_iter = P.max_iter; key.nparts = nparts; key.use_mfma = use_mfma ? 1 : (use_syrk ? 2 : 0);
        BaGraphCache* gc = profiling ? nullptr : ba_graph_cache(P.graph_cache);
        BaGraphEntry* e = gc ? gc->find(key, s) : nullptr;
        if (e && e->exec && !gc->disabled) {
            if (hipGraphLaunch(e->exec, s) == hipSuccess) break;
            (void)hipGetLastError();
            gc->disabled = true;  // fall through to plain launches, now and from here on
        }
        if (e && !e->exec && !gc->disabled && e->seen >= 1) {
            hipStream_t cs = s;
            s = gc->capture_stream;
            hipGraph_t g = nullptr;
            bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                ok = init() == 0;
                for (int it = 0; ok && it < P.max_iter; it++) { normal_equations(it); solve_update(it); }
                ok = (hipStreamEndCapture(s, &g) == hipSuccess) && ok && g;
            }
            s = cs;
            if (ok) ok = hipGraphInstantiate(&e->exec, g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
            if (ok && hipGraphLaunch(e->exec, s) == hipSuccess) break;
            (void)hipGetLastError();
            if (e->exec) { (void)hipGraphExecDestroy(e->exec); e->exec = nullptr; }
            gc->disabled = true;
        }
        if (e) e->seen++;
        int r = init();
        if (r) return r;
        for (int it = 0; it < P.max_iter; it++) { normal_equations(it); solve_update(it); }
        break;
    }
    case 0: { int r = init(); if (r) return r; break; }
    case 1: normal_equations(P.it); break;
    case 2: solve_update(P.it); break;
    case 3: hipLaunchKernelGGL(k_ba_finalize, dim3(1, nw), dim3(64), 0, s, J, P.it); break;
    default: return -4;
    }
    return (int)hipGetLastError();
}
