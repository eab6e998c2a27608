// NLS pose kernels (K11-K13, K15-K18): estimateWorldCameraPose / fcnNLS_t / fcnNLS_Rt (utils/NLS.py:9-33,102-183),
// fcnMSV1_t + fcn2vintercept (utils/MSV.py:8-49,98-142) and the projection helpers of utils/common.py.
//
// One workgroup (16 wavefronts) solves one problem start to finish: every Levenberg-Marquardt iteration is a streaming
// pass over the points (forward-difference Jacobian rows exactly as the reference builds them, dx = 1e-6, float64),
// a wavefront butterfly + LDS reduction of the J^T J / J^T r terms, and a tiny dense solve on one lane.  All iterations
// are fused in one launch: the data (32 B/point) stays in L2/registers, so the kernel is latency bound, not HBM bound.
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
        const double ik = 1.0 / (q2 + FD_STEP * K[3 * k + 2]);
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

// ---------------------------------------------------------------------------------------------------------------
// estimateWorldCameraPose (NLS.py:9-33).  mode 0: fcnNLS_t (3 DoF), mode 1: fcnNLS_Rt (6 DoF).
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, int NLS_THREADS>
__global__ __launch_bounds__(NLS_THREADS) void k_pose(const void* tab, size_t stride)
{
    constexpr int NLS_WAVES = NLS_THREADS / 64;
    const PoseJob J = pjob(tab, stride, blockIdx.x);  // by value: pointers / counts live in SGPRs
    if (J.mode != MODE) return;
    const int n = J.n_ptr ? *J.n_ptr : J.n;
    const int tid = threadIdx.x;
    __shared__ double sh[(MODE == 0 ? 9 : 27) * NLS_WAVES];
    __shared__ double s_x[6];
    __shared__ int s_stop, s_iters;
    double K[9];
    for (int k = 0; k < 9; k++) K[k] = J.K[k];

    if (tid == 0) {
        if (MODE == 0) { for (int k = 0; k < 3; k++) s_x[k] = J.x0[3 + k]; }
        else { for (int k = 0; k < 6; k++) s_x[k] = J.x0[k]; }
        s_stop = 0;
        s_iters = 0;
    }
    __syncthreads();

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
                block_sum_f64<9, NLS_WAVES>(acc, sh);
                if (tid == 0) {
                    double x[3] = {x0, x1, x2};
                    const double r = lm_update<3>(acc, gain, x);
                    s_x[0] = x[0]; s_x[1] = x[1]; s_x[2] = x[2];
                    s_iters = it + 1;
                    if (r < 1e-8) s_stop = 1;
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
                block_sum_f64<27, NLS_WAVES>(acc, sh);
                if (tid == 0) {
                    const double r = lm_update<6>(acc, gain, x);
                    for (int k = 0; k < 6; k++) s_x[k] = x[k];
                    s_iters = it + 1;
                    if (r < 1e-8) s_stop = 1;
                }
            }
            __syncthreads();
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
    block_sum_f64<1, NLS_WAVES>(ss, sh);
    if (tid == 0) {
        for (int k = 0; k < 3; k++) J.t_out[k] = (float)t[k];
        if (J.R_out) for (int k = 0; k < 9; k++) J.R_out[k] = R[k];
        *J.res_out = n > 0 ? sqrt(ss[0] / (2.0 * n)) : 0.0;
        J.info_out[0] = s_iters;
        J.info_out[1] = converged;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// point-wise helpers of utils/common.py (row-vector convention)
// ---------------------------------------------------------------------------------------------------------------
// world2image (common.py:58-64): C = [R; t] @ K (4x3, row-major), out = pscale([pw,1] @ C)
__global__ void k_world2image(const double* C, const double* pw, int n, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double w0 = pw[3 * i], w1 = pw[3 * i + 1], w2 = pw[3 * i + 2];
    const double q0 = w0 * C[0] + w1 * C[3] + w2 * C[6] + C[9];
    const double q1 = w0 * C[1] + w1 * C[4] + w2 * C[7] + C[10];
    const double q2 = w0 * C[2] + w1 * C[5] + w2 * C[8] + C[11];
    out[2 * i] = q0 / q2;
    out[2 * i + 1] = q1 / q2;
}
// image2world (common.py:49-55): Hi = inv([R[0:2]; t] @ K), out = pscale([p,1] @ Hi)
__global__ void k_image2world(const double* Hi, const double* p, int n, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = p[2 * i], y = p[2 * i + 1];
    const double q0 = x * Hi[0] + y * Hi[3] + Hi[6];
    const double q1 = x * Hi[1] + y * Hi[4] + Hi[7];
    const double q2 = x * Hi[2] + y * Hi[5] + Hi[8];
    out[2 * i] = q0 / q2;
    out[2 * i + 1] = q1 / q2;
}
// pixel2uvec (common.py:122-126)
__global__ void k_pixel2uvec(double cx, double cy, double f, const double* p, int n, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r[3];
    uvec_f64(p[2 * i], p[2 * i + 1], cx, cy, f, r);
    out[3 * i] = r[0]; out[3 * i + 1] = r[1]; out[3 * i + 2] = r[2];
}

// ---------------------------------------------------------------------------------------------------------------
// fcn2vintercept (MSV.py:98-142): mean of the pairwise closest-approach points over all C(nf,2) frame pairs.
// A: [nf,3] origins, U: [3,nf,nv] unit directions, out: [nv,3]
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void two_view_point(const double* A, const double* U, int nf, int nv, int i, double* c)
{
    double sx = 0, sy = 0, sz = 0, Bx = 0, By = 0, Bz = 0;
    int npairs = 0;
    for (int j = 0; j < nf; j++) { Bx += A[3 * j]; By += A[3 * j + 1]; Bz += A[3 * j + 2]; }
    for (int j = 0; j < nf; j++) {
        const double ux = U[(0 * nf + j) * nv + i], uy = U[(1 * nf + j) * nv + i], uz = U[(2 * nf + j) * nv + i];
        for (int k = j + 1; k < nf; k++) {
            const double vx = U[(0 * nf + k) * nv + i], vy = U[(1 * nf + k) * nv + i], vz = U[(2 * nf + k) * nv + i];
            const double dAx = A[3 * j] - A[3 * k], dAy = A[3 * j + 1] - A[3 * k + 1], dAz = A[3 * j + 2] - A[3 * k + 2];
            const double d = ux * vx + uy * vy + uz * vz;
            const double e = ux * dAx + uy * dAy + uz * dAz;
            const double f = vx * dAx + vy * dAy + vz * dAz;
            const double g = 1 - d * d;
            const double s1 = (d * f - e) / g, t1 = (f - d * e) / g;
            sx += t1 * vx + s1 * ux;
            sy += t1 * vy + s1 * uy;
            sz += t1 * vz + s1 * uz;
            npairs++;
        }
    }
    const double den = 2.0 * npairs, m = (double)(nf - 1);
    c[0] = (sx + Bx * m) / den;
    c[1] = (sy + By * m) / den;
    c[2] = (sz + Bz * m) / den;
}

__global__ void k_two_view(const double* A, const double* U, int nf, int nv, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    double c[3];
    two_view_point(A, U, nf, nv, i, c);
    out[3 * i] = c[0]; out[3 * i + 1] = c[1]; out[3 * i + 2] = c[2];
}

// fcnNvintercept (MSV.py:146-175): least-squares intersection of the nf rays of every track,
//   C0 = inv(sum_f (I - u u^T)) (sum_f (I - u u^T) A_f)
__global__ void k_n_view(const double* A, const double* U, int nf, int nv, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int f = 0; f < nf; f++) {
        const double u[3] = {U[(0 * nf + f) * nv + i], U[(1 * nf + f) * nv + i], U[(2 * nf + f) * nv + i]};
        const double a[3] = {A[3 * f], A[3 * f + 1], A[3 * f + 2]};
        for (int r = 0; r < 3; r++) {
            double acc = 0.0;
            for (int c = 0; c < 3; c++) {
                const double v = (r == c ? 1.0 : 0.0) - u[r] * u[c];
                S[r * 3 + c] += v;
                acc += v * a[c];
            }
            b[r] += acc;
        }
    }
    solve_dense<3>(S, b);
    out[3 * i] = b[0]; out[3 * i + 1] = b[1]; out[3 * i + 2] = b[2];
}

// ---------------------------------------------------------------------------------------------------------------
// fcnMSV1_t (MSV.py:8-49): LM over the last camera translation; every iteration re-triangulates all points.
// ---------------------------------------------------------------------------------------------------------------
#define MSV_THREADS 1024
__global__ __launch_bounds__(MSV_THREADS) void k_msv1(MsvJob J)
{
    constexpr int NLS_THREADS = MSV_THREADS, NLS_WAVES = MSV_THREADS / 64;
    const int tid = threadIdx.x;
    const int ng = J.ng_ptr ? *J.ng_ptr : J.ng;
    const int nf = J.nf;
    __shared__ double sh[9 * NLS_WAVES];
    __shared__ double s_x[3], s_A[3 * 16];
    __shared__ int s_stop, s_iters;
    double K[9];
    for (int k = 0; k < 9; k++) K[k] = J.K[k];

    // unit rays U[:, j, g] = pixel2uvec(K, P[0:2, vg, j])  (MSV.py:15-17); P is float32 [5, N0, nhist]
    for (int q = tid; q < ng * nf; q += NLS_THREADS) {
        const int g = q % ng, j = q / ng;
        const int id = J.ids ? J.ids[g] : g;
        const float pu = J.P[((size_t)0 * J.N0 + id) * J.nhist + j], pv = J.P[((size_t)1 * J.N0 + id) * J.nhist + j];
        double r[3];
        if (J.f32_rays) uvec_f32(pu, pv, (float)K[6], (float)K[7], (float)K[0], r);  // K and P float32 -> numpy works in float32
        else uvec_f64((double)pu, (double)pv, K[6], K[7], K[0], r);
        J.U[((size_t)0 * nf + j) * ng + g] = r[0];
        J.U[((size_t)1 * nf + j) * ng + g] = r[1];
        J.U[((size_t)2 * nf + j) * ng + g] = r[2];
    }
    if (tid == 0) {
        // u0 = B[0,0:3] - B[:nf,0:3] ; x = [0,0,1] - u0[nf-2]   (MSV.py:18-19), B float32 [nhist,14]
        for (int j = 0; j < nf; j++)
            for (int c = 0; c < 3; c++) s_A[3 * j + c] = (double)(float)(J.B[c] - J.B[14 * j + c]);
        const double e[3] = {0, 0, 1};
        for (int c = 0; c < 3; c++) s_x[c] = e[c] - s_A[3 * (nf - 2) + c];
        s_stop = 0;
        s_iters = 0;
    }
    __threadfence_block();
    __syncthreads();

    int converged = 0;
    for (int it = 0; it < J.max_iter && ng > 0; it++) {
        const double x0 = s_x[0], x1 = s_x[1], x2 = s_x[2];
        double A[3 * 16];
        for (int j = 0; j < nf - 1; j++)
            for (int c = 0; c < 3; c++) A[3 * j + c] = s_A[3 * j + c];
        A[3 * (nf - 1)] = -x0; A[3 * (nf - 1) + 1] = -x1; A[3 * (nf - 1) + 2] = -x2;  // vstack(u0[:-1], -x)  (MSV.py:29)
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = tid; g < ng; g += NLS_THREADS) {
            const int id = J.ids ? J.ids[g] : g;
            double c[3];
            two_view_point(A, J.U, nf, ng, g, c);
            const double b0 = c[0] + x0, b1 = c[1] + x1, b2 = c[2] + x2;
            J.b0[3 * g] = b0; J.b0[3 * g + 1] = b1; J.b0[3 * g + 2] = b2;
            double u, v, ju[3], jv[3];
            fd_rows_t(K, b0, b1, b2, u, v, ju, jv);
            const double zu = (double)J.P[((size_t)0 * J.N0 + id) * J.nhist + (nf - 1)];
            const double zv = (double)J.P[((size_t)1 * J.N0 + id) * J.nhist + (nf - 1)];
            accumulate<3>(acc, ju, jv, zu - u, zv - v);
        }
        block_sum_f64<9, NLS_WAVES>(acc, sh);
        if (tid == 0) {
            double x[3] = {x0, x1, x2};
            const double r = lm_update<3>(acc, 1.0, x);  // no step ramp (MSV.py:36)
            s_x[0] = x[0]; s_x[1] = x[1]; s_x[2] = x[2];
            s_iters = it + 1;
            if (r < 1e-8) s_stop = 1;
        }
        __syncthreads();
        if (s_stop) { converged = 1; break; }
    }
    if (tid == 0) {
        for (int c = 0; c < 3; c++) J.x_out[c] = (float)s_x[c];
        J.info_out[0] = s_iters;
        J.info_out[1] = converged;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------------------
void vh_launch_pose(const void* tab, size_t stride, int batch, int mode, int max_n, hipStream_t s)
{
    // translation fit: 4 wavefronts keep up to 4096 points in registers; an LM iteration is then bound by the block
    // reduction + the serial 3x3 update, which 16 wavefronts only make longer (74 us vs 50 us per 2000-point fit)
    if (mode == 0 && max_n <= 4096) hipLaunchKernelGGL((k_pose<0, 256>), dim3(batch), dim3(256), 0, s, tab, stride);
    else if (mode == 0) hipLaunchKernelGGL((k_pose<0, 1024>), dim3(batch), dim3(1024), 0, s, tab, stride);
    else hipLaunchKernelGGL((k_pose<1, 256>), dim3(batch), dim3(256), 0, s, tab, stride);
}
void vh_launch_world2image(const double* C, const double* pw, int n, double* out, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_world2image, dim3((n + 255) / 256), dim3(256), 0, s, C, pw, n, out);
}
void vh_launch_image2world(const double* Hi, const double* p, int n, double* out, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_image2world, dim3((n + 255) / 256), dim3(256), 0, s, Hi, p, n, out);
}
void vh_launch_pixel2uvec(double cx, double cy, double f, const double* p, int n, double* out, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_pixel2uvec, dim3((n + 255) / 256), dim3(256), 0, s, cx, cy, f, p, n, out);
}
void vh_launch_two_view(const double* A, const double* U, int nf, int nv, double* out, hipStream_t s)
{
    if (nv > 0) hipLaunchKernelGGL(k_two_view, dim3((nv + 255) / 256), dim3(256), 0, s, A, U, nf, nv, out);
}
void vh_launch_n_view(const double* A, const double* U, int nf, int nv, double* out, hipStream_t s)
{
    if (nv > 0) hipLaunchKernelGGL(k_n_view, dim3((nv + 255) / 256), dim3(256), 0, s, A, U, nf, nv, out);
}
void vh_launch_msv1(const MsvJob& job, hipStream_t s) { hipLaunchKernelGGL(k_msv1, dim3(1), dim3(MSV_THREADS), 0, s, job); }
