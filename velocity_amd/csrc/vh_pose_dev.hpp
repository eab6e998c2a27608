// Device-side pose solver (estimateWorldCameraPose / fcnNLS_t / fcnNLS_Rt, utils/NLS.py:9-33,102-183) shared by vh_nls.hip (k_pose) and
// vh_session.hip (k_sess_frame: bookkeeping + pose + records of one frame in a single launch).
#pragma once
#include "vh_nls.hpp"

#define FD_STEP 1e-6

// block-wide sum of NV doubles; the result is valid in THREAD 0 only (the lane that does the dense solve).
// sh: [NV * NLS_WAVES]
template <int NV, int NLS_WAVES>
__device__ void block_sum_f64(double* v, double* sh)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = vh_wave_sum_f64(v[k]);
    __syncthreads();
    if (lane == 0)
        for (int k = 0; k < NV; k++) sh[k * NLS_WAVES + wave] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            double s = 0.0;
            for (int q = 0; q < NLS_WAVES; q++) s += sh[k * NLS_WAVES + q];
            v[k] = s;
        }
    }
}

// block-wide sum of NV doubles, valid in EVERY thread after ONE barrier: the per-wave sums go to the buffer `sh` ([NV * NLS_WAVES]; the caller
// alternates between two buffers), every thread adds them in the same order.  What follows a reduction in the LM loop (the dense solve, the
// state update, the stop rule) is then computed redundantly by all threads from identical inputs: no broadcast, no second and third barrier.
template <int NV, int NLS_WAVES>
__device__ __forceinline__ void block_sum_f64_all(double* v, double* sh)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = vh_wave_sum_f64(v[k]);
    if (lane == 0)
        for (int k = 0; k < NV; k++) sh[k * NLS_WAVES + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NLS_WAVES; q++) s += sh[k * NLS_WAVES + q];
        v[k] = s;
    }
}

// The 9 sums of the 3-DoF normal equations (6 entries of J^T J, 3 of J^T r) of one wavefront, TRANSPOSING while folding: v_permlane32_swap puts the upper
// half-wave of one value beside the lower half-wave of another, so one add folds TWO values 64 -> 32 lanes, v_permlane16_swap does the same for the
// 16-lane rows, and the last four steps are row rotations (DPP).  5 + 3 swap-folds and 3 x 4 rotate-adds (60 instructions) instead of 9 x 6 butterflies
// through ds_bpermute (two per step and value: 160 instructions and six LDS-crossbar round trips); the totals end up in lanes 0 / 16 / 32 / 48, which store
// them.  One barrier, then every thread adds the per-wave sums in the same order (block_sum_f64_all's contract).
__device__ __forceinline__ double pose_u2d(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); }
__device__ __forceinline__ double fold32_pair_f64(double a, double b)  // lanes 0..31: a[l] + a[l+32], lanes 32..63: b[l-32] + b[l]
{
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    return pose_u2d(lo[0], hi[0]) + pose_u2d(lo[1], hi[1]);
}
__device__ __forceinline__ double fold16_pair_f64(double a, double b)  // rows 0/2: a[row] + a[row+1], rows 1/3: b[row-1] + b[row]
{
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    return pose_u2d(lo[0], hi[0]) + pose_u2d(lo[1], hi[1]);
}
template <int N>
__device__ __forceinline__ double row_ror_add_f64(double v)  // v[l] + v[(l + N) mod 16 within the 16-lane row]
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x120 + N, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x120 + N, 0xf, 0xf, false);
    return v + pose_u2d(lo, hi);
}
__device__ __forceinline__ double row_sum_f64(double v)
{
    v = row_ror_add_f64<8>(v);
    v = row_ror_add_f64<4>(v);
    v = row_ror_add_f64<2>(v);
    return row_ror_add_f64<1>(v);
}
template <int NLS_WAVES>
__device__ __forceinline__ void block_sum9_f64_all(double* v, double* sh)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double r0 = fold32_pair_f64(v[0], v[1]), r1 = fold32_pair_f64(v[2], v[3]), r2 = fold32_pair_f64(v[4], v[5]), r3 = fold32_pair_f64(v[6], v[7]);
    const double r4 = fold32_pair_f64(v[8], v[8]);
    // rows of q0: v0 v2 v1 v3; of q1: v4 v6 v5 v7; of q2: v8 (four times)
    const double q0 = row_sum_f64(fold16_pair_f64(r0, r1)), q1 = row_sum_f64(fold16_pair_f64(r2, r3)), q2 = row_sum_f64(fold16_pair_f64(r4, r4));
    if ((lane & 15) == 0) {
        const int row = lane >> 4, i0 = ((row & 1) << 1) | (row >> 1);
        sh[i0 * NLS_WAVES + wave] = q0;
        sh[(4 + i0) * NLS_WAVES + wave] = q1;
        if (row == 0) sh[8 * NLS_WAVES + wave] = q2;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; k++) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NLS_WAVES; q++) s += sh[k * NLS_WAVES + q];
        v[k] = s;
    }
}

// uv of camera-frame point b:  pscale(b @ K)   (fzK, NLS.py:71-78)
__device__ __forceinline__ void project_cam(const double* K, double b0, double b1, double b2, double& u, double& v)
{
    const double q0 = b0 * K[0] + b1 * K[3] + b2 * K[6];
    const double q1 = b0 * K[1] + b1 * K[4] + b2 * K[7];
    const double q2 = b0 * K[2] + b1 * K[5] + b2 * K[8];
    u = q0 / q2;
    v = q1 / q2;
}

// pixel2uvec (common.py:122-126) in the dtype numpy would use
__device__ __forceinline__ void uvec_f64(double pu, double pv, double cx, double cy, double f, double* r)
{
    const double a = pu - cx, b = pv - cy;
    const double nrm = sqrt(a * a + b * b + f * f);
    r[0] = a / nrm; r[1] = b / nrm; r[2] = f / nrm;
}
__device__ __forceinline__ void uvec_f32(float pu, float pv, float cx, float cy, float f, double* r)
{
    const float a = __fsub_rn(pu, cx), b = __fsub_rn(pv, cy);
    const float nrm = vh_sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)), __fmul_rn(f, f)));
    r[0] = (double)__fdiv_rn(a, nrm); r[1] = (double)__fdiv_rn(b, nrm); r[2] = (double)__fdiv_rn(f, nrm);
}


// Residual point + forward-difference rows of the translation Jacobian (fcnNLS_t / fcnMSV1_t: b0 + dx e_k re-projected,
// NLS.py:119-120).  (b + dx e_k) @ K = q + dx K[k,:], so the three perturbed projections reuse q; one reciprocal per
// projection.  Same forward-difference values as the reference up to float64 rounding (~1e-16 rel).
__device__ __forceinline__ void fd_rows_t(const double* K, double b0, double b1, double b2, double& u, double& v, double* ju, double* jv)
{
    const double q0 = b0 * K[0] + b1 * K[3] + b2 * K[6];
    const double q1 = b0 * K[1] + b1 * K[4] + b2 * K[7];
    const double q2 = b0 * K[2] + b1 * K[5] + b2 * K[8];
    const double iq = 1.0 / q2;
    u = q0 * iq;
    v = q1 * iq;
    const double inv_dx = 1.0 / FD_STEP;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // K's third column is (0, 0, 1) for a pinhole camera (images.py:148-151): q2 + dx * 0 is q2 itself, so the reciprocal of the
        // unperturbed projection is reused (same bits); a general K takes the division
        const double ik = K[3 * k + 2] == 0.0 ? iq : 1.0 / (q2 + FD_STEP * K[3 * k + 2]);
        ju[k] = ((q0 + FD_STEP * K[3 * k]) * ik - u) * inv_dx;
        jv[k] = ((q1 + FD_STEP * K[3 * k + 1]) * ik - v) * inv_dx;
    }
}

__device__ void rpy2dcm(const double* rpy, double* C)  // transforms.py:7-23
{
    const double sr = sin(rpy[0]), cr = cos(rpy[0]), sp = sin(rpy[1]), cp = cos(rpy[1]), sy = sin(rpy[2]), cy = cos(rpy[2]);
    C[0] = cp * cy; C[1] = sr * sp * cy - cr * sy; C[2] = cr * sp * cy + sr * sy;
    C[3] = cp * sy; C[4] = sr * sp * sy + cr * cy; C[5] = cr * sp * sy - sr * cy;
    C[6] = -sp;     C[7] = sr * cp;                C[8] = cr * cp;
}

// x = inv(A) b for a small dense system (Gauss-Jordan with partial pivoting, like LAPACK getrf/getri)
template <int N>
__device__ void solve_dense(double* A /* N*N row-major, destroyed */, double* b /* in: rhs, out: solution */)
{
    for (int c = 0; c < N; c++) {
        int piv = c;
        double best = fabs(A[c * N + c]);
        for (int r = c + 1; r < N; r++)
            if (fabs(A[r * N + c]) > best) { best = fabs(A[r * N + c]); piv = r; }
        if (piv != c) {
            for (int k = 0; k < N; k++) { const double t = A[c * N + k]; A[c * N + k] = A[piv * N + k]; A[piv * N + k] = t; }
            const double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        const double inv = 1.0 / A[c * N + c];
        for (int r = 0; r < N; r++) {
            if (r == c) continue;
            const double f = A[r * N + c] * inv;
            for (int k = c; k < N; k++) A[r * N + k] -= f * A[c * N + k];
            b[r] -= f * b[c];
        }
    }
    for (int c = 0; c < N; c++) b[c] /= A[c * N + c];
}

__device__ __forceinline__ const PoseJob& pjob(const void* tab, size_t stride, int b)
{
    return *reinterpret_cast<const PoseJob*>(reinterpret_cast<const char*>(tab) + (size_t)b * stride);
}

// Accumulate the NP x NP normal equations from one measurement pair (u,v) with forward-difference rows.
template <int NP>
__device__ __forceinline__ void accumulate(double* acc, const double* ju, const double* jv, double ru, double rv)
{
    int q = 0;
#pragma unroll
    for (int a = 0; a < NP; a++)
#pragma unroll
        for (int b = a; b < NP; b++) acc[q++] += ju[a] * ju[b] + jv[a] * jv[b];
#pragma unroll
    for (int a = 0; a < NP; a++) acc[q++] += ju[a] * ru + jv[a] * rv;
}

// LM update on lane 0 from the reduced sums; returns rms(delta) and writes delta*gain into x
template <int NP>
__device__ double lm_update(const double* acc, double gain, double* x)
{
    double A[NP * NP], g[NP];
    int q = 0;
    for (int a = 0; a < NP; a++)
        for (int b = a; b < NP; b++) { A[a * NP + b] = acc[q]; A[b * NP + a] = acc[q]; q++; }
    for (int a = 0; a < NP; a++) { A[a * NP + a] += 1.0; g[a] = acc[q++]; }  // constant +I damping (NLS.py:115,154)
    solve_dense<NP>(A, g);
    double ss = 0.0;
    for (int a = 0; a < NP; a++) { const double d = g[a] * gain; x[a] += d; ss += d * d; }
    return sqrt(ss / NP);
}

// The 3-DoF update.  A = J^T J + I is symmetric positive definite, so delta = A^-1 g needs no pivoting: adjugate / determinant, ONE division on the
// dependency chain where the pivoted elimination above has six (every thread runs this between two reductions of the LM loop: its latency is paid per
// iteration).  Returns sum(d^2); the caller's stop rule  rms(d) = sqrt(sum / 3) < 1e-8  (NLS.py:125) is evaluated as  sum < POSE_STOP3, the smallest double
// whose rms is not below 1e-8 (sqrt and the division by 3 are monotone and correctly rounded: the two tests agree for EVERY double, checked by bisection).
#define POSE_STOP3 0x1.59e05f1e2674dp-52
__device__ __forceinline__ double lm_update3(const double* acc, double gain, double* x)
{
    const double a = acc[0] + 1.0, b = acc[1], c = acc[2], d = acc[3] + 1.0, e = acc[4], f = acc[5] + 1.0;  // constant +I damping (NLS.py:115)
    const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
    const double D = a * f - c * c, E = b * c - a * e, F = a * d - b * b;
    const double id = gain / (a * A + b * B + c * C);
    const double d0 = (A * acc[6] + B * acc[7] + C * acc[8]) * id;
    const double d1 = (B * acc[6] + D * acc[7] + E * acc[8]) * id;
    const double d2 = (C * acc[6] + E * acc[7] + F * acc[8]) * id;
    x[0] += d0; x[1] += d1; x[2] += d2;
    return d0 * d0 + d1 * d1 + d2 * d2;
}

// ---------------------------------------------------------------------------------------------------------------
// estimateWorldCameraPose (NLS.py:9-33).  mode 0: fcnNLS_t (3 DoF), mode 1: fcnNLS_Rt (6 DoF).
// ---------------------------------------------------------------------------------------------------------------
// One workgroup of NLS_THREADS threads solves the problem start to finish (called by k_pose and, fused between the bookkeeping halves of a
// frame, by the tracker session's k_sess_frame).  Ends with a barrier-free tail: the caller synchronises before reading the outputs.
template <int MODE, int NLS_THREADS>
__device__ __forceinline__ void pose_solve(const PoseJob& J)
{
    constexpr int NLS_WAVES = NLS_THREADS / 64;
    const int n = J.n_ptr ? *J.n_ptr : J.n;
    const int tid = threadIdx.x;
    __shared__ double sh2[2][(MODE == 0 ? 9 : 27) * NLS_WAVES];  // two reduction buffers, alternated: one barrier per reduction
    double* sh = sh2[0];
    int par = 0;
    // the state, the stop flag and the iteration count are wave-uniform REGISTERS: every thread runs the (tiny) solve itself
    double s_x[6];
    int s_stop = 0, s_iters = 0;
    double K[9];
    for (int k = 0; k < 9; k++) K[k] = J.K[k];
    if (MODE == 0) { for (int k = 0; k < 3; k++) s_x[k] = J.x0[3 + k]; }
    else { for (int k = 0; k < 6; k++) s_x[k] = J.x0[k]; }

    // the points of this thread stay in registers across the LM iterations (n <= PPT * NLS_THREADS), so an iteration
    // is arithmetic + one reduction, not a chain of dependent global loads
    constexpr int PPT = 4096 / NLS_THREADS;  // register-cached points per thread: up to 4096 points per problem
    const bool cached = MODE == 0 && n <= PPT * NLS_THREADS;
    double cw[PPT][3], cz[PPT][2];
    if (cached) {
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = tid + q * NLS_THREADS;
            if (i < n) {
                const int ip = J.p_sel ? J.p_sel[i] : i, iw = J.pw_sel ? J.pw_sel[i] : i;
                cw[q][0] = J.pw[3 * iw]; cw[q][1] = J.pw[3 * iw + 1]; cw[q][2] = J.pw[3 * iw + 2];
                cz[q][0] = (double)J.p[2 * ip]; cz[q][1] = (double)J.p[2 * ip + 1];
            }
        }
    }
    const int max_iter = 30;
    int converged = 0;
    if (n > 0) {
        for (int it = 0; it < max_iter; it++) {
            double gain = (it + 1) * 0.2;
            gain = gain * gain;
            if (gain > 1.0) gain = 1.0;
            if (MODE == 0) {
                const double x0 = s_x[0], x1 = s_x[1], x2 = s_x[2];
                double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (cached) {
#pragma unroll
                    for (int q = 0; q < PPT; q++) {
                        if (tid + q * NLS_THREADS < n) {
                            const double b0 = cw[q][0] + x0, b1 = cw[q][1] + x1, b2 = cw[q][2] + x2;
                            double u, v, ju[3], jv[3];
                            fd_rows_t(K, b0, b1, b2, u, v, ju, jv);
                            accumulate<3>(acc, ju, jv, cz[q][0] - u, cz[q][1] - v);
                        }
                    }
                } else
                for (int i = tid; i < n; i += NLS_THREADS) {
                    const int ip = J.p_sel ? J.p_sel[i] : i, iw = J.pw_sel ? J.pw_sel[i] : i;
                    const double b0 = J.pw[3 * iw] + x0, b1 = J.pw[3 * iw + 1] + x1, b2 = J.pw[3 * iw + 2] + x2;
                    double u, v, ju[3], jv[3];
                    fd_rows_t(K, b0, b1, b2, u, v, ju, jv);
                    accumulate<3>(acc, ju, jv, (double)J.p[2 * ip] - u, (double)J.p[2 * ip + 1] - v);
                }
                block_sum9_f64_all<NLS_WAVES>(acc, sh2[par]);
                par ^= 1;
                {
                    double x[3] = {x0, x1, x2};
                    const double ssd = lm_update3(acc, gain, x);
                    s_x[0] = x[0]; s_x[1] = x[1]; s_x[2] = x[2];
                    s_iters = it + 1;
                    if (ssd < POSE_STOP3) s_stop = 1;
                }
            } else {
                double x[6], R0[9], Rk[3][9];
                for (int k = 0; k < 6; k++) x[k] = s_x[k];
                rpy2dcm(x, R0);
                for (int k = 0; k < 3; k++) {
                    double a[3] = {x[0], x[1], x[2]};
                    a[k] += FD_STEP;
                    rpy2dcm(a, Rk[k]);
                }
                double acc[27];
                for (int k = 0; k < 27; k++) acc[k] = 0.0;
                for (int i = tid; i < n; i += NLS_THREADS) {
                    const int ip = J.p_sel ? J.p_sel[i] : i, iw = J.pw_sel ? J.pw_sel[i] : i;
                    const double w0 = J.pw[3 * iw], w1 = J.pw[3 * iw + 1], w2 = J.pw[3 * iw + 2];
                    const double a0 = w0 * R0[0] + w1 * R0[3] + w2 * R0[6];
                    const double a1 = w0 * R0[1] + w1 * R0[4] + w2 * R0[7];
                    const double a2 = w0 * R0[2] + w1 * R0[5] + w2 * R0[8];
                    double u, v, ju[6], jv[6], uk, vk;
                    project_cam(K, a0 + x[3], a1 + x[4], a2 + x[5], u, v);
                    for (int k = 0; k < 3; k++) {
                        const double c0 = w0 * Rk[k][0] + w1 * Rk[k][3] + w2 * Rk[k][6];
                        const double c1 = w0 * Rk[k][1] + w1 * Rk[k][4] + w2 * Rk[k][7];
                        const double c2 = w0 * Rk[k][2] + w1 * Rk[k][5] + w2 * Rk[k][8];
                        project_cam(K, c0 + x[3], c1 + x[4], c2 + x[5], uk, vk);
                        ju[k] = (uk - u) / FD_STEP; jv[k] = (vk - v) / FD_STEP;
                    }
                    project_cam(K, a0 + (x[3] + FD_STEP), a1 + x[4], a2 + x[5], uk, vk); ju[3] = (uk - u) / FD_STEP; jv[3] = (vk - v) / FD_STEP;
                    project_cam(K, a0 + x[3], a1 + (x[4] + FD_STEP), a2 + x[5], uk, vk); ju[4] = (uk - u) / FD_STEP; jv[4] = (vk - v) / FD_STEP;
                    project_cam(K, a0 + x[3], a1 + x[4], a2 + (x[5] + FD_STEP), uk, vk); ju[5] = (uk - u) / FD_STEP; jv[5] = (vk - v) / FD_STEP;
                    accumulate<6>(acc, ju, jv, (double)J.p[2 * ip] - u, (double)J.p[2 * ip + 1] - v);
                }
                block_sum_f64_all<27, NLS_WAVES>(acc, sh2[par]);
                par ^= 1;
                {
                    const double r = lm_update<6>(acc, gain, x);
                    for (int k = 0; k < 6; k++) s_x[k] = x[k];
                    s_iters = it + 1;
                    if (r < 1e-8) s_stop = 1;
                }
            }
            if (s_stop) { converged = 1; break; }
        }
    }

    // outputs: t (float32, NLS.py:129,181), R, then p_proj = world2image(K, R, t, p3) and rms(p - p_proj) (NLS.py:31-32)
    double R[9], t[3];
    if (MODE == 0) {
        for (int k = 0; k < 9; k++) R[k] = J.R[k];
        for (int k = 0; k < 3; k++) t[k] = (double)(float)s_x[k];
    } else {
        double Rd[9];
        rpy2dcm(s_x, Rd);
        for (int k = 0; k < 9; k++) R[k] = (double)(float)Rd[k];
        for (int k = 0; k < 3; k++) t[k] = (double)(float)s_x[3 + k];
    }
    double C[12];  // camMatrix = [R; t] @ K  (4x3)
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C[r * 3 + c] = R[r * 3] * K[c] + R[r * 3 + 1] * K[3 + c] + R[r * 3 + 2] * K[6 + c];
    for (int c = 0; c < 3; c++) C[9 + c] = t[0] * K[c] + t[1] * K[3 + c] + t[2] * K[6 + c];
    double ss[1] = {0.0};
    for (int i = tid; i < n; i += NLS_THREADS) {
        const int ip = J.p_sel ? J.p_sel[i] : i, iw = J.pw_sel ? J.pw_sel[i] : i;
        const double w0 = J.pw[3 * iw], w1 = J.pw[3 * iw + 1], w2 = J.pw[3 * iw + 2];
        const double q0 = w0 * C[0] + w1 * C[3] + w2 * C[6] + C[9];
        const double q1 = w0 * C[1] + w1 * C[4] + w2 * C[7] + C[10];
        const double q2 = w0 * C[2] + w1 * C[5] + w2 * C[8] + C[11];
        const double u = q0 / q2, v = q1 / q2;
        if (J.p_proj) { J.p_proj[2 * i] = u; J.p_proj[2 * i + 1] = v; }
        const double du = (double)J.p[2 * ip] - u, dv = (double)J.p[2 * ip + 1] - v;
        ss[0] += du * du + dv * dv;
    }
    sh = sh2[par];
    block_sum_f64_all<1, NLS_WAVES>(ss, sh);
    if (tid == 0) {
        for (int k = 0; k < 3; k++) J.t_out[k] = (float)t[k];
        if (J.R_out) for (int k = 0; k < 9; k++) J.R_out[k] = R[k];
        *J.res_out = n > 0 ? sqrt(ss[0] / (2.0 * n)) : 0.0;
        J.info_out[0] = s_iters;
        J.info_out[1] = converged;
    }
}

