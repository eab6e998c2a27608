// NLS pose kernels (K11-K13, K15-K18): estimateWorldCameraPose / fcnNLS_t / fcnNLS_Rt (utils/NLS.py:9-33,102-183),
// fcnMSV1_t + fcn2vintercept (utils/MSV.py:8-49,98-142) and the projection helpers of utils/common.py.
//
// One workgroup (16 wavefronts) solves one problem start to finish: every Levenberg-Marquardt iteration is a streaming
// pass over the points (forward-difference Jacobian rows exactly as the reference builds them, dx = 1e-6, float64),
// a wavefront butterfly + LDS reduction of the J^T J / J^T r terms, and a tiny dense solve on one lane.  All iterations
// are fused in one launch: the data (32 B/point) stays in L2/registers, so the kernel is latency bound, not HBM bound.
#include "vh_nls.hpp"

#include "vh_pose_dev.hpp"

template <int MODE, int NLS_THREADS>
__global__ __launch_bounds__(NLS_THREADS) void k_pose(const void* tab, size_t stride)
{
    const PoseJob J = pjob(tab, stride, blockIdx.x);  // by value: pointers / counts live in SGPRs
    if (J.mode != MODE) return;
    pose_solve<MODE, NLS_THREADS>(J);
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

// the same in float32 when K and p are float32 (numpy then computes in float32; uvec's sum of squares and square root included)
__global__ void k_pixel2uvec_f32(float cx, float cy, float f, const float* p, int n, float* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r[3];
    uvec_f32(p[2 * i], p[2 * i + 1], cx, cy, f, r);  // float32 arithmetic, results exactly representable in float32
    out[3 * i] = (float)r[0]; out[3 * i + 1] = (float)r[1]; out[3 * i + 2] = (float)r[2];
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
__device__ __forceinline__ void msv1_body(const MsvJob& J)
{
    constexpr int NLS_THREADS = MSV_THREADS, NLS_WAVES = MSV_THREADS / 64;
    const int tid = threadIdx.x;
    const int ng = J.ng_ptr ? *J.ng_ptr : J.ng;
    const int nf = J.nf;
    __shared__ double sh[9 * NLS_WAVES];
    __shared__ double s_x[3];
    extern __shared__ double s_A[];  // [nf][3] ray origins; row nf-1 is rewritten every iteration (any number of frames: 24 B of LDS each)
    __shared__ int s_stop, s_iters;
    double K[9];
    for (int k = 0; k < 9; k++) K[k] = J.K[k];

    // unit rays U[:, j, g] = pixel2uvec(K, P[0:2, vg, j])  (MSV.py:15-17); P is float32 [5, N0, nhist]
    for (int q = tid; q < ng * nf; q += NLS_THREADS) {
        const int g = q % ng, j = q / ng;
        const int id = J.ids ? J.ids[g] : g;
        const float pu = J.P[(size_t)id * J.P_ts + (size_t)j * J.P_fs], pv = J.P[J.P_rs + (size_t)id * J.P_ts + (size_t)j * J.P_fs];
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
        for (int c = 0; c < 3; c++) s_A[3 * (nf - 1) + c] = -s_x[c];  // vstack(u0[:-1], -x)  (MSV.py:29); u0's own last row is never used
        s_stop = 0;
        s_iters = 0;
    }
    __threadfence_block();
    __syncthreads();

    int converged = 0;
    for (int it = 0; it < J.max_iter && ng > 0; it++) {
        const double x0 = s_x[0], x1 = s_x[1], x2 = s_x[2];
        const double* A = s_A;  // wave-uniform LDS reads (broadcast)
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = tid; g < ng; g += NLS_THREADS) {
            const int id = J.ids ? J.ids[g] : g;
            double c[3];
            two_view_point(A, J.U, nf, ng, g, c);
            const double b0 = c[0] + x0, b1 = c[1] + x1, b2 = c[2] + x2;
            J.b0[3 * g] = b0; J.b0[3 * g + 1] = b1; J.b0[3 * g + 2] = b2;
            double u, v, ju[3], jv[3];
            fd_rows_t(K, b0, b1, b2, u, v, ju, jv);
            const double zu = (double)J.P[(size_t)id * J.P_ts + (size_t)(nf - 1) * J.P_fs];
            const double zv = (double)J.P[J.P_rs + (size_t)id * J.P_ts + (size_t)(nf - 1) * J.P_fs];
            accumulate<3>(acc, ju, jv, zu - u, zv - v);
        }
        block_sum_f64<9, NLS_WAVES>(acc, sh);
        if (tid == 0) {
            double x[3] = {x0, x1, x2};
            const double r = lm_update<3>(acc, 1.0, x);  // no step ramp (MSV.py:36)
            s_x[0] = x[0]; s_x[1] = x[1]; s_x[2] = x[2];
            s_A[3 * (nf - 1)] = -x[0]; s_A[3 * (nf - 1) + 1] = -x[1]; s_A[3 * (nf - 1) + 2] = -x[2];
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

__global__ __launch_bounds__(MSV_THREADS) void k_msv1(MsvJob J) { msv1_body(J); }

// One workgroup per video stream of a session, ONE launch for all of them (round 4 launched a one-workgroup kernel per stream: 256 dependent launches
// of ~0.2 ms each at the frame where every stream of a batch re-triangulates).  `tab` points at the MsvJob inside stream 0's device record, records are
// `stride` bytes apart, and the stream's own frame counter sits `frame_off` bytes from its job: only the streams that are AT their MSV frame run
// (vidExample.py:155 `if i == msvFrame`), the other workgroups leave at once.
__global__ __launch_bounds__(MSV_THREADS) void k_msv1_tab(const void* tab, size_t stride, ptrdiff_t frame_off, int fire_frame)
{
    const char* base = reinterpret_cast<const char*>(tab) + (size_t)blockIdx.x * stride;
    if (*reinterpret_cast<const int*>(base + frame_off) != fire_frame) return;
    msv1_body(*reinterpret_cast<const MsvJob*>(base));
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
void vh_launch_pixel2uvec_f32(float cx, float cy, float f, const float* p, int n, float* out, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_pixel2uvec_f32, dim3((n + 255) / 256), dim3(256), 0, s, cx, cy, f, p, n, out);
}
void vh_launch_two_view(const double* A, const double* U, int nf, int nv, double* out, hipStream_t s)
{
    if (nv > 0) hipLaunchKernelGGL(k_two_view, dim3((nv + 255) / 256), dim3(256), 0, s, A, U, nf, nv, out);
}
void vh_launch_n_view(const double* A, const double* U, int nf, int nv, double* out, hipStream_t s)
{
    if (nv > 0) hipLaunchKernelGGL(k_n_view, dim3((nv + 255) / 256), dim3(256), 0, s, A, U, nf, nv, out);
}
void vh_launch_msv1(const MsvJob& job, hipStream_t s)
{
    hipLaunchKernelGGL(k_msv1, dim3(1), dim3(MSV_THREADS), sizeof(double) * 3 * (size_t)job.nf, s, job);
}
void vh_launch_msv1_tab(const void* tab, size_t stride, ptrdiff_t frame_off, int fire_frame, int nf, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_msv1_tab, dim3(batch), dim3(MSV_THREADS), sizeof(double) * 3 * (size_t)nf, s, tab, stride, frame_off, fire_frame);
}
