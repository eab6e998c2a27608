// Bundle adjustment (K14): fcnNLS_batch (utils/NLS.py:186-250) -- dense Levenberg-Marquardt over all tie points and
// cameras 1..nc with the constant +I damping, step 0.9, <= 10 iterations.
//
// The reference materialises the dense forward-difference J^T (nx x nz) and inverts J^T J + I.  J has 9 non-zeros per
// row (3 for the point, 6 for the camera), so here the forward-difference entries are computed exactly as the reference
// defines them (same dx = 1e-6 perturbations of x, same re-projection) but kept compact, and the damped normal equations
// are solved through the point-block Schur complement -- algebraically the same delta as the dense inverse:
//     H = [[U  W],[W^T V]] + I,   S = V + I - W^T (U+I)^-1 W,   S dc = gc - W^T (U+I)^-1 gp,   dp = (U+I)^-1 (gp - W dc)
// Everything stays on the device for all iterations (the convergence test only sets a device flag).
#include "vh_ba.hpp"

#define BA_FD 1e-6
#define BA_THREADS 256
#define BA_EPT 64  // Schur entries owned by one thread per pass

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
    u = q0 / q2;
    v = q1 / q2;
}

// camera rotation matrices: R(rpy) and the three forward-difference neighbours R(rpy + dx e_k)  (NLS.py:206-216,228-233)
__global__ void k_ba_cams(BaJob J)
{
    if (*J.done) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;  // camera index 0..nc (0 = fixed identity camera)
    if (c > J.nc) return;
    double* out = J.camR + (size_t)c * 36;
    if (c == 0) {
        for (int q = 0; q < 4; q++)
            for (int k = 0; k < 9; k++) out[q * 9 + k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double* rpy = J.x + 3 * J.nt + 3 * J.nc + 3 * (c - 1);
    ba_rpy2dcm(rpy, out);
    for (int k = 0; k < 3; k++) {
        double a[3] = {rpy[0], rpy[1], rpy[2]};
        a[k] += BA_FD;
        ba_rpy2dcm(a, out + 9 * (k + 1));
    }
}

// residual and compact forward-difference Jacobian of every measurement pair (camera c, track i)
__global__ __launch_bounds__(BA_THREADS) void k_ba_jac(BaJob J)
{
    if (*J.done) return;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int nt = J.nt, nf = J.nc + 1;
    double ss = 0.0;
    if (m < nt * nf) {
        const int c = m / nt, i = m - c * nt;
        double K[9];
        for (int k = 0; k < 9; k++) K[k] = J.K[k];
        const double* R = J.camR + (size_t)c * 36;
        double w[3] = {J.x[3 * i], J.x[3 * i + 1], J.x[3 * i + 2]};
        double t[3] = {0, 0, 0};
        if (c > 0) for (int k = 0; k < 3; k++) t[k] = J.x[3 * nt + 3 * (c - 1) + k];
        double u, v, uk, vk;
        ba_project(K, R, w, t, u, v);
        const double ru = J.z[m] - u, rv = J.z[(size_t)nt * nf + m] - v;  // z = [all u | all v], camera-major (NLS.py:198-199)
        J.r[2 * (size_t)m] = ru;
        J.r[2 * (size_t)m + 1] = rv;
        ss = ru * ru + rv * rv;
        double* Jp = J.Jp + 6 * (size_t)m;
        for (int k = 0; k < 3; k++) {  // point coordinates
            double wk[3] = {w[0], w[1], w[2]};
            wk[k] += BA_FD;
            ba_project(K, R, wk, t, uk, vk);
            Jp[k] = (uk - u) / BA_FD;
            Jp[3 + k] = (vk - v) / BA_FD;
        }
        double* Jc = J.Jc + 12 * (size_t)m;
        if (c > 0) {
            for (int k = 0; k < 3; k++) {  // camera position
                double tk[3] = {t[0], t[1], t[2]};
                tk[k] += BA_FD;
                ba_project(K, R, w, tk, uk, vk);
                Jc[k] = (uk - u) / BA_FD;
                Jc[6 + k] = (vk - v) / BA_FD;
            }
            for (int k = 0; k < 3; k++) {  // camera roll / pitch / yaw
                ba_project(K, R + 9 * (k + 1), w, t, uk, vk);
                Jc[3 + k] = (uk - u) / BA_FD;
                Jc[9 + k] = (vk - v) / BA_FD;
            }
        } else {
            for (int k = 0; k < 12; k++) Jc[k] = 0.0;
        }
    }
    // sum of squared residuals of this iteration (trace only)
    ss = vh_wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0 && ss != 0.0) atomicAdd(J.acc, ss);
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
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = 6 * nc, tid = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sW = reinterpret_cast<double*>(smem);  // [3][nq]
    double* sY = sW + 3 * nq;                      // [3][nq]
    double* sJc = sY + 3 * nq;                     // [nc][12]
    double* sTp = sJc + 12 * nc;                   // [3] tp_i, [3..] scratch
    const int chunk = (nt + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(nt, i0 + chunk);
    const long long nent = (long long)nq * nq;
    const long long ebase = (long long)pass * BA_THREADS * BA_EPT;
    double accS[BA_EPT];
#pragma unroll
    for (int e = 0; e < BA_EPT; e++) accS[e] = 0.0;
    double accR = 0.0;  // reduced rhs entry `tid` (first pass only, tid < nq)

    for (int i = i0; i < i1; i++) {
        __syncthreads();
        // thread 0..: U_i and gp_i over all cameras (tiny) -- done redundantly by the first wave's lane 0
        if (tid == 0) {
            double U[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, gp[3] = {0, 0, 0};  // +I damping (NLS.py:220)
            for (int c = 0; c <= nc; c++) {
                const size_t m = (size_t)c * nt + i;
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
        // stage the camera Jacobians of this point
        for (int q = tid; q < 12 * nc; q += BA_THREADS) {
            const int c = q / 12 + 1, k = q - (c - 1) * 12;
            sJc[q] = J.Jc[12 * ((size_t)c * nt + i) + k];
        }
        __syncthreads();
        // W_i [3][nq] and Y_i = U_i^-1 W_i
        for (int q = tid; q < nq; q += BA_THREADS) {
            const int c = q / 6, k = q - 6 * c;  // camera c+1, parameter k (0..2 pos, 3..5 rpy)
            const double* Jp = J.Jp + 6 * ((size_t)(c + 1) * nt + i);
            const double ju = sJc[12 * c + k], jv = sJc[12 * c + 6 + k];
            const double w0 = Jp[0] * ju + Jp[3] * jv, w1 = Jp[1] * ju + Jp[4] * jv, w2 = Jp[2] * ju + Jp[5] * jv;
            sW[q] = w0; sW[nq + q] = w1; sW[2 * nq + q] = w2;
            const double* Ui = sTp + 3;
            const double y0 = Ui[0] * w0 + Ui[1] * w1 + Ui[2] * w2, y1 = Ui[3] * w0 + Ui[4] * w1 + Ui[5] * w2, y2 = Ui[6] * w0 + Ui[7] * w1 + Ui[8] * w2;
            sY[q] = y0; sY[nq + q] = y1; sY[2 * nq + q] = y2;
            if (pass == 0) {
                double* Yg = J.Y + ((size_t)i * nq + q) * 3;
                Yg[0] = y0; Yg[1] = y1; Yg[2] = y2;
                // reduced rhs: gc - W^T tp
                const size_t m = (size_t)(c + 1) * nt + i;
                accR += ju * J.r[2 * m] + jv * J.r[2 * m + 1] - (w0 * sTp[0] + w1 * sTp[1] + w2 * sTp[2]);
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
        // accR holds the contributions of rhs entries q = tid, tid + 256, ... ; nq <= 256 is required by the launcher
        if (tid < nq) J.Rpart[(size_t)blockIdx.x * nq + tid] = accR;
    }
}

// Schur stage 2: reduce the partials, add the +I damping, solve S dc = rhs with Gauss-Jordan + partial pivoting
__global__ __launch_bounds__(BA_THREADS) void k_ba_solve(BaJob J, int nparts)
{
    if (*J.done) return;
    const int nq = 6 * J.nc, tid = threadIdx.x, ld = nq + 1;
    double* A = J.Sfull;  // [nq][nq+1] augmented
    const long long nent = (long long)nq * nq;
    for (long long e = tid; e < nent; e += BA_THREADS) {
        double s = 0.0;
        for (int p = 0; p < nparts; p++) s += J.Spart[(size_t)p * nent + e];
        const int a = (int)(e / nq), b = (int)(e - (long long)a * nq);
        A[(size_t)a * ld + b] = s + (a == b ? 1.0 : 0.0);
    }
    for (int q = tid; q < nq; q += BA_THREADS) {
        double s = 0.0;
        for (int p = 0; p < nparts; p++) s += J.Rpart[(size_t)p * nq + q];
        A[(size_t)q * ld + nq] = s;
    }
    __shared__ int s_piv;
    __shared__ double s_inv;
    __syncthreads();
    for (int c = 0; c < nq; c++) {
        if (tid == 0) {
            int piv = c;
            double best = fabs(A[(size_t)c * ld + c]);
            for (int r = c + 1; r < nq; r++) {
                const double v = fabs(A[(size_t)r * ld + c]);
                if (v > best) { best = v; piv = r; }
            }
            s_piv = piv;
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv != c)
            for (int k = tid; k <= nq; k += BA_THREADS) {
                const double t = A[(size_t)c * ld + k];
                A[(size_t)c * ld + k] = A[(size_t)piv * ld + k];
                A[(size_t)piv * ld + k] = t;
            }
        __syncthreads();
        if (tid == 0) s_inv = 1.0 / A[(size_t)c * ld + c];
        __syncthreads();
        const double inv = s_inv;
        // eliminate column c from every other row; thread t handles rows t, t + 256, ...
        for (int r = tid; r < nq; r += BA_THREADS) {
            if (r == c) continue;
            const double f = A[(size_t)r * ld + c] * inv;
            if (f != 0.0)
                for (int k = c; k <= nq; k++) A[(size_t)r * ld + k] -= f * A[(size_t)c * ld + k];
        }
        __syncthreads();
    }
    for (int q = tid; q < nq; q += BA_THREADS) J.dc[q] = A[(size_t)q * ld + nq] / A[(size_t)q * ld + q];
}

// back-substitution dp = tp - Y dc, update x += 0.9 delta, rms(delta) and the stop flag (NLS.py:235-240)
__global__ __launch_bounds__(BA_THREADS) void k_ba_update(BaJob J, int it)
{
    if (*J.done) return;
    const int nt = J.nt, nc = J.nc, nq = 6 * nc, tid = threadIdx.x;
    __shared__ double sh[BA_THREADS / 64];
    double ss = 0.0;
    for (int i = blockIdx.x * BA_THREADS + tid; i < nt; i += gridDim.x * BA_THREADS) {
        double d[3] = {J.tp[3 * (size_t)i], J.tp[3 * (size_t)i + 1], J.tp[3 * (size_t)i + 2]};
        const double* Y = J.Y + (size_t)i * nq * 3;
        for (int q = 0; q < nq; q++) {
            const double dq = J.dc[q];
            d[0] -= Y[3 * q] * dq; d[1] -= Y[3 * q + 1] * dq; d[2] -= Y[3 * q + 2] * dq;
        }
        for (int k = 0; k < 3; k++) {
            const double dl = d[k] * 0.9;
            J.x[3 * (size_t)i + k] += dl;
            ss += dl * dl;
        }
    }
    if (blockIdx.x == 0)
        for (int q = tid; q < nq; q += BA_THREADS) {
            const int c = q / 6, k = q - 6 * c;
            const double dl = J.dc[q] * 0.9;
            // state layout: [points | camera positions | camera rpy] (NLS.py:203)
            const size_t idx = k < 3 ? (size_t)3 * nt + 3 * c + k : (size_t)3 * nt + 3 * nc + 3 * c + (k - 3);
            J.x[idx] += dl;
            ss += dl * dl;
        }
    ss = vh_wave_sum_f64(ss);
    if ((tid & 63) == 0) sh[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < BA_THREADS / 64; k++) s += sh[k];
        atomicAdd(J.acc + 1, s);
        __threadfence();
        const unsigned prev = atomicAdd(J.ticket, 1u);
        if (prev == gridDim.x - 1) {  // last block: finish the iteration record
            const double nz = 2.0 * nt * (nc + 1), nx = 3.0 * nt + 6.0 * nc;
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

// ---------------------------------------------------------------------------------------------------------------
size_t vh_ba_workspace_bytes(int nt, int nc, int nparts)
{
    const size_t nf = nc + 1, nq = 6 * (size_t)nc, m = (size_t)nt * nf;
    size_t b = 0;
    auto add = [&](size_t n) { b += (n * sizeof(double) + 255) / 256 * 256; };
    add(36 * nf); add(2 * m); add(6 * m); add(12 * m); add(3 * (size_t)nt); add(3 * nq * nt); add(nparts * nq * nq); add(nparts * nq);
    add(nq * (nq + 1)); add(nq); add(4); add(32);
    return b + 1024;
}

int vh_ba_run(const BaProblem& P, hipStream_t s)
{
    const int nt = P.nt, nc = P.nc, nq = 6 * nc;
    if (nq > BA_THREADS) return -3;  // reduced rhs ownership (one thread per entry) needs 6 nc <= 256
    const int nparts = P.nparts;
    BaJob J;
    J.nt = nt; J.nc = nc;
    for (int k = 0; k < 9; k++) J.K[k] = P.K[k];
    J.z = P.z; J.x = P.x; J.trace = P.trace; J.info = P.info;
    char* w = reinterpret_cast<char*>(P.workspace);
    auto take = [&](size_t n) { double* p = reinterpret_cast<double*>(w); w += (n * sizeof(double) + 255) / 256 * 256; return p; };
    const size_t nf = nc + 1, m = (size_t)nt * nf;
    J.camR = take(36 * nf); J.r = take(2 * m); J.Jp = take(6 * m); J.Jc = take(12 * m); J.tp = take(3 * (size_t)nt); J.Y = take(3 * (size_t)nq * nt);
    J.Spart = take((size_t)nparts * nq * nq); J.Rpart = take((size_t)nparts * nq); J.Sfull = take((size_t)nq * (nq + 1)); J.dc = take(nq);
    J.acc = take(4);
    double* flags = take(32);
    J.done = reinterpret_cast<int*>(flags);
    J.ticket = reinterpret_cast<unsigned*>(flags) + 4;
    hipError_t e = hipMemsetAsync(J.acc, 0, 4 * sizeof(double), s);
    if (e == hipSuccess) e = hipMemsetAsync(flags, 0, 32 * sizeof(double), s);
    if (e == hipSuccess) e = hipMemsetAsync(P.info, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    const long long nent = (long long)nq * nq;
    const int npass = (int)((nent + (long long)BA_THREADS * BA_EPT - 1) / ((long long)BA_THREADS * BA_EPT));
    const size_t lds = sizeof(double) * (size_t)(6 * nq + 12 * nc + 16);
    const int nmeas = nt * (nc + 1);
    const int upd_blocks = (nt + BA_THREADS - 1) / BA_THREADS;
    for (int it = 0; it < P.max_iter; it++) {
        hipLaunchKernelGGL(k_ba_cams, dim3((nc + 1 + 63) / 64), dim3(64), 0, s, J);
        hipLaunchKernelGGL(k_ba_jac, dim3((nmeas + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, s, J);
        for (int pass = 0; pass < npass; pass++) {
            // later passes overwrite Spart entries of their own range only
            hipLaunchKernelGGL(k_ba_points, dim3(nparts), dim3(BA_THREADS), lds, s, J, pass);
        }
        hipLaunchKernelGGL(k_ba_solve, dim3(1), dim3(BA_THREADS), 0, s, J, nparts);
        hipLaunchKernelGGL(k_ba_update, dim3(upd_blocks), dim3(BA_THREADS), 0, s, J, it);
    }
    return (int)hipGetLastError();
}
