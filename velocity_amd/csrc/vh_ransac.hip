// Deterministic RANSAC affine + least-squares refit (K6): stands in for cv2.estimateAffine2D(method=RANSAC)
// (utils/KLT.py:116,127).  OpenCV's RNG sequence cannot be reproduced without its generator, so the
// sample of hypothesis h comes from a counter-based hash; all VH_RANSAC_ITERS hypotheses are scored in parallel (one
// wavefront each) and a single thread then replays OpenCV's sequential "best so far + adaptive iteration count" rule
// over the scores, which gives exactly the result of the sequential algorithm.  The refit sums are exact int64
// fixed-point sums, so the result does not depend on reduction order.
#include <atomic>
#include "vh_kernels.hpp"
#include "vh_glue_dev.hpp"

#define RANSAC_THRESH2 9.0f
#define RANSAC_CONF 0.99
#define RANSAC_SEED 0x2545F491u
#define RANSAC_HEAD 16  // hypotheses scored inside the compaction kernel

__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t ransac_draw(uint32_t hyp, uint32_t k, uint32_t attempt, uint32_t m)
{
    const uint32_t h = mix32(RANSAC_SEED ^ mix32(hyp * 0x9E3779B9u + k * 0x7F4A7C15u + attempt * 0x94D049BBu + 1u));
    return (uint32_t)(((uint64_t)h * m) >> 32);
}

// log(x), x > 0, from plain IEEE arithmetic only (bit-identical on host and device)
__device__ double det_log(double x)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) { m = __dmul_rn(m, 0.5); e += 1; }
    const double s = __ddiv_rn(__dsub_rn(m, 1.0), __dadd_rn(m, 1.0)), s2 = __dmul_rn(s, s);
    // 1 / (2k + 1): compile-time constants (correctly rounded, exactly what the division returns at run time) -- 13 dependent float64
    // divisions per call were most of the single-thread selection replay
    constexpr double R[13] = {1.0,        1.0 / 3.0,  1.0 / 5.0,  1.0 / 7.0,  1.0 / 9.0,  1.0 / 11.0, 1.0 / 13.0,
                              1.0 / 15.0, 1.0 / 17.0, 1.0 / 19.0, 1.0 / 21.0, 1.0 / 23.0, 1.0 / 25.0};
    double acc = 0.0;
#pragma unroll
    for (int k = 12; k >= 0; k--) acc = __dadd_rn(__dmul_rn(acc, s2), R[k]);
    return __dadd_rn(__dmul_rn((double)e, 0.6931471805599453), __dmul_rn(__dmul_rn(2.0, s), acc));
}

__device__ int ransac_update_iters(double conf, double ep, int max_iters)
{
    double num = __dsub_rn(1.0, conf);
    if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
    const double wgt = __dsub_rn(1.0, ep);
    double denom = __dsub_rn(1.0, __dmul_rn(__dmul_rn(wgt, wgt), wgt));
    if (denom < 2.2250738585072014e-308) return 0;
    num = det_log(num);
    denom = det_log(denom);
    if (denom >= 0 || -num >= __dmul_rn((double)max_iters, -denom)) return max_iters;
    return __double2int_rn(__ddiv_rn(num, denom));
}

__device__ __forceinline__ bool collinear(const double* a, const double* b, const double* c)
{
    const double dx1 = __dsub_rn(b[0], a[0]), dy1 = __dsub_rn(b[1], a[1]), dx2 = __dsub_rn(c[0], a[0]), dy2 = __dsub_rn(c[1], a[1]);
    const double lhs = fabs(__dsub_rn(__dmul_rn(dx1, dy2), __dmul_rn(dy1, dx2)));
    const double rhs = __dmul_rn(1.1920929e-07, __dadd_rn(__dadd_rn(__dadd_rn(fabs(dx1), fabs(dy1)), fabs(dx2)), fabs(dy2)));
    return lhs <= rhs;
}

// hypothesis `hyp` over the m compacted pairs; returns false when the sample is degenerate
__device__ bool ransac_hypothesis(const float* from, const float* to, const int* idx, int m, uint32_t hyp, double* M)
{
    int id[3];
    for (int k = 0; k < 3; k++) {
        bool ok = false;
        for (uint32_t a = 0; a < 16 && !ok; a++) {
            id[k] = (int)ransac_draw(hyp, (uint32_t)k, a, (uint32_t)m);
            ok = true;
            for (int q = 0; q < k; q++) ok = ok && (id[q] != id[k]);
        }
        if (!ok) return false;
    }
    double f[3][2], t[3][2];
    for (int k = 0; k < 3; k++) {
        const int i = idx[id[k]];
        f[k][0] = from[2 * i]; f[k][1] = from[2 * i + 1];
        t[k][0] = to[2 * i];   t[k][1] = to[2 * i + 1];
    }
    if (collinear(f[0], f[1], f[2]) || collinear(t[0], t[1], t[2])) return false;
    const double ax = __dsub_rn(f[0][0], f[2][0]), ay = __dsub_rn(f[0][1], f[2][1]);
    const double bx = __dsub_rn(f[1][0], f[2][0]), by = __dsub_rn(f[1][1], f[2][1]);
    const double det = __dsub_rn(__dmul_rn(ax, by), __dmul_rn(bx, ay));
    if (det == 0.0) return false;
    for (int r = 0; r < 2; r++) {
        const double u0 = __dsub_rn(t[0][r], t[2][r]), u1 = __dsub_rn(t[1][r], t[2][r]);
        const double a = __ddiv_rn(__dsub_rn(__dmul_rn(u0, by), __dmul_rn(u1, ay)), det);
        const double b = __ddiv_rn(__dsub_rn(__dmul_rn(ax, u1), __dmul_rn(bx, u0)), det);
        M[3 * r] = a;
        M[3 * r + 1] = b;
        M[3 * r + 2] = __dsub_rn(__dsub_rn(t[2][r], __dmul_rn(a, f[2][0])), __dmul_rn(b, f[2][1]));
    }
    return true;
}

__device__ __forceinline__ bool is_inlier(const double* M, float x, float y, float u, float v)
{
    const double ex = __dsub_rn(__dadd_rn(__dadd_rn(__dmul_rn(M[0], x), __dmul_rn(M[1], y)), M[2]), (double)u);
    const double ey = __dsub_rn(__dadd_rn(__dadd_rn(__dmul_rn(M[3], x), __dmul_rn(M[4], y)), M[5]), (double)v);
    const float e = (float)__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey));
    return e <= RANSAC_THRESH2;
}

__device__ __forceinline__ const RansacJob& rjob(const void* tab, size_t stride, int b)
{
    return *reinterpret_cast<const RansacJob*>(reinterpret_cast<const char*>(tab) + (size_t)b * stride);
}

// block-wide sum of NV int64 values; result valid in every thread
template <int NV>
__device__ void block_sum_i64(long long* v, long long* sh /* [NV * 4] */)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = vh_wave_sum_i64(v[k]);
    __syncthreads();
    if (lane == 0)
        for (int k = 0; k < NV; k++) sh[k * 4 + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = sh[k * 4] + sh[k * 4 + 1] + sh[k * 4 + 2] + sh[k * 4 + 3];
}

// inliers of model M among the m compacted pairs, counted by one wavefront (one 16-byte load per pair)
__device__ __forceinline__ int score_pairs(const float4* pairs, int m, const double* M, int lane)
{
    int c = 0;
    for (int k = lane; k < m; k += 64) {
        const float4 q = pairs[k];
        c += is_inlier(M, q.x, q.y, q.z, q.w) ? 1 : 0;
    }
    return vh_wave_sum_i32(c);
}

// ---- 1. compaction of the valid pairs (order preserving) + the first RANSAC_HEAD hypotheses ---------------------
// One 1024-thread block per stream: the whole head (16 hypotheses) is scored in one round, one wavefront each.
__global__ __launch_bounds__(1024) void k_ransac_compact(const void* tab, size_t stride)
{
    const RansacJob J = rjob(tab, stride, blockIdx.x);  // by value: fields live in SGPRs
    const int n = J.n_ptr ? *J.n_ptr : J.n;
    __shared__ int wcount[16];
    __shared__ int base;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c = 0; c < n; c += 1024) {
        const int i = c + tid;
        const bool f = i < n && J.valid[i] != 0;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f) {
            const float2 a = reinterpret_cast<const float2*>(J.from)[i], b = reinterpret_cast<const float2*>(J.to)[i];
            q = make_float4(a.x, a.y, b.x, b.y);
        }
        const unsigned long long bal = __ballot(f);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int cw = wcount[w];
            off += w < wave ? cw : 0;
            tot += cw;
        }
        if (f) { J.idx[off + pre] = i; J.pairs[off + pre] = q; }
        __syncthreads();
        if (tid == 0) base += tot;
        __syncthreads();
    }
    if (tid == 0) *J.m_out = base;
    __syncthreads();  // also makes this block's J.idx / J.pairs stores visible to all of its waves
    // Score the first RANSAC_HEAD hypotheses here and replay the sequential rule over them: the adaptive iteration
    // count can only shrink afterwards, so hypotheses >= *bound can never be reached and need not be scored.
    const int m = base;
    int bound = VH_RANSAC_ITERS;
    if (m >= 3 && m > J.min_valid) {
        static_assert(RANSAC_HEAD == 16, "one wavefront per head hypothesis");
        {
            double M[6];
            int c = 0;
            if (ransac_hypothesis(J.from, J.to, J.idx, m, (uint32_t)wave, M)) c = score_pairs(J.pairs, m, M, lane);
            if (lane == 0) { J.counts[wave] = c; wcount[wave] = c; }  // LDS copy: the replay below must not walk global memory
        }
        __syncthreads();
        if (tid == 0) {
            int best_count = 0, niters = VH_RANSAC_ITERS;
            for (int it = 0; it < RANSAC_HEAD && it < niters; it++) {
                const int c = wcount[it];
                if (c > max(best_count, 2)) {
                    best_count = c;
                    niters = ransac_update_iters(RANSAC_CONF, __ddiv_rn((double)(m - c), (double)m), niters);
                }
            }
            bound = niters;
        }
    }
    if (tid == 0) *J.bound = bound;
}

// ---- 2. score every hypothesis: one wavefront per hypothesis ----------------------------------------------------
__global__ __launch_bounds__(256) void k_ransac_score(const void* tab, size_t stride)
{
    const RansacJob J = rjob(tab, stride, blockIdx.y);  // by value: fields live in SGPRs
    const int m = *J.m_out;
    const int hyp = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (hyp >= VH_RANSAC_ITERS || hyp < RANSAC_HEAD || hyp >= *J.bound) return;  // head already scored; tail unreachable
    if (m < 3 || m <= J.min_valid) return;
    double M[6];
    int c = 0;
    if (ransac_hypothesis(J.from, J.to, J.idx, m, (uint32_t)hyp, M)) c = score_pairs(J.pairs, m, M, lane);
    if (lane == 0) J.counts[hyp] = c;
}

// ---- 3. sequential selection rule, inlier mask, least-squares refit ---------------------------------------------
__global__ __launch_bounds__(256) void k_ransac_select(const void* tab, size_t stride)
{
    const RansacJob J = rjob(tab, stride, blockIdx.x);  // by value: fields live in SGPRs
    const int n = J.n_ptr ? *J.n_ptr : J.n;
    const int m = *J.m_out;
    const int tid = threadIdx.x;
    __shared__ int s_best, s_count;
    __shared__ double s_M[6];
    __shared__ long long s_red[9 * 4];
    // the scores of every reachable hypothesis in ONE coalesced round trip: the single-thread replay below walks them with data-dependent
    // control flow, and from global memory every step of it was a full L2 latency (most of this kernel's 14 us)
    __shared__ int s_counts[VH_RANSAC_ITERS];
    {
        // the replay reads entry `it` while it < (current) niters: the whole head (niters may only drop below 16 at its end) and then up to the bound
        const int nb = min(max(*J.bound, RANSAC_HEAD), VH_RANSAC_ITERS);
        for (int k = tid; k < nb; k += 256) s_counts[k] = J.counts[k];
    }
    __syncthreads();
    if (tid == 0) {
        int best = -1, best_count = 0;
        if (m >= 3 && m > J.min_valid) {
            int niters = VH_RANSAC_ITERS;
            for (int it = 0; it < niters; it++) {
                const int c = s_counts[it];
                if (c > max(best_count, 2)) {
                    best = it;
                    best_count = c;
                    niters = ransac_update_iters(RANSAC_CONF, __ddiv_rn((double)(m - c), (double)m), niters);
                }
            }
        }
        s_best = best;
        s_count = best_count;
        if (best >= 0) {
            double M[6];
            ransac_hypothesis(J.from, J.to, J.idx, m, (uint32_t)best, M);
            for (int k = 0; k < 6; k++) s_M[k] = M[k];
        }
    }
    __syncthreads();
    const int best = s_best;
    if (best < 0) {
        for (int i = tid; i < n; i += 256) {
            J.inl[i] = 0;
            if (J.gate_valid) J.valid[i] = 0;
        }
        if (tid == 0) *J.status = 0;
        return;
    }
    double M[6];
    for (int k = 0; k < 6; k++) M[k] = s_M[k];

    // inlier mask + first fixed-point pass (means), 2^-32 resolution
    long long s1[4] = {0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        bool in = false;
        if (J.valid[i]) {
            const float x = J.from[2 * i], y = J.from[2 * i + 1], u = J.to[2 * i], v = J.to[2 * i + 1];
            in = is_inlier(M, x, y, u, v);
            if (in) {
                s1[0] += vh_fixq((double)x, 32); s1[1] += vh_fixq((double)y, 32);
                s1[2] += vh_fixq((double)u, 32); s1[3] += vh_fixq((double)v, 32);
            }
        }
        J.inl[i] = in ? 1 : 0;
    }
    block_sum_i64<4>(s1, s_red);
    const double cnt = (double)s_count;
    const double mx = __ddiv_rn(ldexp((double)s1[0], -32), cnt), my = __ddiv_rn(ldexp((double)s1[1], -32), cnt);
    const double mu = __ddiv_rn(ldexp((double)s1[2], -32), cnt), mv = __ddiv_rn(ldexp((double)s1[3], -32), cnt);
    __syncthreads();  // J.inl written by this block is re-read below (same thread reads its own entries)

    long long q[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        if (J.inl[i]) {
            const double x = __dsub_rn((double)J.from[2 * i], mx), y = __dsub_rn((double)J.from[2 * i + 1], my);
            const double u = __dsub_rn((double)J.to[2 * i], mu), v = __dsub_rn((double)J.to[2 * i + 1], mv);
            q[0] += vh_fixq(__dmul_rn(x, x), 20); q[1] += vh_fixq(__dmul_rn(x, y), 20); q[2] += vh_fixq(__dmul_rn(y, y), 20);
            q[3] += vh_fixq(__dmul_rn(x, u), 20); q[4] += vh_fixq(__dmul_rn(y, u), 20);
            q[5] += vh_fixq(__dmul_rn(x, v), 20); q[6] += vh_fixq(__dmul_rn(y, v), 20);
        }
    }
    block_sum_i64<7>(q, s_red);
    if (J.gate_valid)
        for (int i = tid; i < n; i += 256) J.valid[i] = J.inl[i];
    if (tid == 0) {
        const double Sxx = ldexp((double)q[0], -20), Sxy = ldexp((double)q[1], -20), Syy = ldexp((double)q[2], -20);
        const double Sxu = ldexp((double)q[3], -20), Syu = ldexp((double)q[4], -20);
        const double Sxv = ldexp((double)q[5], -20), Syv = ldexp((double)q[6], -20);
        const double det = __dsub_rn(__dmul_rn(Sxx, Syy), __dmul_rn(Sxy, Sxy));
        const double tr = __dadd_rn(Sxx, Syy);
        if (s_count >= 3 && det > __dmul_rn(__dmul_rn(1e-9, tr), tr) && det > 0) {
            const double a = __ddiv_rn(__dsub_rn(__dmul_rn(Sxu, Syy), __dmul_rn(Syu, Sxy)), det);
            const double b = __ddiv_rn(__dsub_rn(__dmul_rn(Sxx, Syu), __dmul_rn(Sxy, Sxu)), det);
            const double d = __ddiv_rn(__dsub_rn(__dmul_rn(Sxv, Syy), __dmul_rn(Syv, Sxy)), det);
            const double e = __ddiv_rn(__dsub_rn(__dmul_rn(Sxx, Syv), __dmul_rn(Sxy, Sxv)), det);
            M[0] = a; M[1] = b; M[2] = __dsub_rn(__dsub_rn(mu, __dmul_rn(a, mx)), __dmul_rn(b, my));
            M[3] = d; M[4] = e; M[5] = __dsub_rn(__dsub_rn(mv, __dmul_rn(d, mx)), __dmul_rn(e, my));
        }
        for (int k = 0; k < 6; k++) J.M[k] = M[k];
        *J.status = 1;
    }
}

// ---- latency path: the three stages in ONE 512-thread workgroup per stream, the compacted pairs resident in LDS -------------------------------
// With few streams in flight the three launches above are a chain of ~13 dependent global-memory round trips (job -> count -> points -> index
// list -> sample -> pairs -> scores -> ...), about 1 us each, for a few microseconds of arithmetic.  Here the compacted (from, to) pairs and
// the scores live in LDS, so after the first read of the points nothing waits on global memory.  Same arithmetic, same integers: bit-identical
// to the three-kernel path (tests/test_gpu_klt.py runs both).  512 threads: a 1024-thread version is capped at 128 VGPRs and spills 516 B / lane.
#define RANSAC_FUSED_MAX 3072  // pairs held in LDS: 3072 x (16 + 4) B + 2000 x 4 B of counts = 69 440 B dynamic (+ ~100 B static)
__device__ bool ransac_hypothesis_lds(const float4* pairs, int m, uint32_t hyp, double* M)
{
    int id[3];
    for (int k = 0; k < 3; k++) {
        bool ok = false;
        for (uint32_t a = 0; a < 16 && !ok; a++) {
            id[k] = (int)ransac_draw(hyp, (uint32_t)k, a, (uint32_t)m);
            ok = true;
            for (int q = 0; q < k; q++) ok = ok && (id[q] != id[k]);
        }
        if (!ok) return false;
    }
    double f[3][2], t[3][2];
    for (int k = 0; k < 3; k++) {
        const float4 q = pairs[id[k]];
        f[k][0] = q.x; f[k][1] = q.y; t[k][0] = q.z; t[k][1] = q.w;
    }
    if (collinear(f[0], f[1], f[2]) || collinear(t[0], t[1], t[2])) return false;
    const double ax = __dsub_rn(f[0][0], f[2][0]), ay = __dsub_rn(f[0][1], f[2][1]);
    const double bx = __dsub_rn(f[1][0], f[2][0]), by = __dsub_rn(f[1][1], f[2][1]);
    const double det = __dsub_rn(__dmul_rn(ax, by), __dmul_rn(bx, ay));
    if (det == 0.0) return false;
    for (int r = 0; r < 2; r++) {
        const double u0 = __dsub_rn(t[0][r], t[2][r]), u1 = __dsub_rn(t[1][r], t[2][r]);
        const double a = __ddiv_rn(__dsub_rn(__dmul_rn(u0, by), __dmul_rn(u1, ay)), det);
        const double b = __ddiv_rn(__dsub_rn(__dmul_rn(ax, u1), __dmul_rn(bx, u0)), det);
        M[3 * r] = a;
        M[3 * r + 1] = b;
        M[3 * r + 2] = __dsub_rn(__dsub_rn(t[2][r], __dmul_rn(a, f[2][0])), __dmul_rn(b, f[2][1]));
    }
    return true;
}

__device__ __forceinline__ void ransac_fused_body(const void* tab, size_t stride)
{
    const RansacJob J = rjob(tab, stride, blockIdx.x);
    const int n = J.n_ptr ? *J.n_ptr : J.n;  // <= RANSAC_FUSED_MAX (the launcher routes larger problems to the three-kernel path)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_pairs = reinterpret_cast<float4*>(smem);                            // [RANSAC_FUSED_MAX] compacted pairs
    int* s_idx = reinterpret_cast<int*>(s_pairs + RANSAC_FUSED_MAX);              // [RANSAC_FUSED_MAX] their point indices
    int* s_counts = s_idx + RANSAC_FUSED_MAX;                                     // [VH_RANSAC_ITERS]
    __shared__ int wcount[8], base, s_bound, s_best, s_count;
    __shared__ double s_M[6];
    __shared__ long long s_red[7 * 8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) base = 0;
    // every chunk's loads are issued before the first barrier (round 6): the compaction was a chain of one global round trip PER 512-pair chunk (four at
    // 2000 pairs: ~6 us of a 21 us single-stream launch); the chunks themselves are then compacted from registers in the same order as before
    constexpr int NCH = RANSAC_FUSED_MAX / 512;
    float4 qv[NCH];
    unsigned fbits = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) qv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n > 0) {
        // unconditional loads at a clamped index (a branch around a load serialises it behind the flag it depends on): flags and pairs of all chunks
        // travel together
        uint8_t fl[NCH];
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            if (512 * j >= n) break;  // (uniform)
            const int ic = min(512 * j + tid, n - 1);
            fl[j] = J.valid[ic];
            const float2 a = reinterpret_cast<const float2*>(J.from)[ic], b = reinterpret_cast<const float2*>(J.to)[ic];
            qv[j] = make_float4(a.x, a.y, b.x, b.y);
        }
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            if (512 * j >= n) break;
            if (512 * j + tid < n && fl[j] != 0) fbits |= 1u << j;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        if (512 * j >= n) break;  // (uniform)
        const int i = 512 * j + tid;
        const bool f = (fbits >> j) & 1u;
        const unsigned long long bal = __ballot(f);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int cw = wcount[w];
            off += w < wave ? cw : 0;
            tot += cw;
        }
        if (f) { s_idx[off + pre] = i; s_pairs[off + pre] = qv[j]; }
        __syncthreads();
        if (tid == 0) base += tot;
        __syncthreads();
    }
    const int m = base;
    if (tid == 0) { *J.m_out = m; s_bound = VH_RANSAC_ITERS; }
    __syncthreads();
    const bool run = m >= 3 && m > J.min_valid;
    // Eight hypotheses per round (one wavefront each); after EVERY round thread 0 continues the sequential selection rule (cv::RANSAC's adaptive iteration
    // count, as k_ransac_select) over the hypotheses scored so far and publishes the count it has shrunk to: the loop ends as soon as every reachable
    // hypothesis is scored.  (Round 5 scored a fixed head of 16 before it looked: on a clean scene hypothesis 0 already ends the search, and the second
    // round was 4.5 of the launch's 20 us -- in-kernel stamps, round 6.)  Same hypotheses, same counts, same rule: bit-identical results.
    int r_best = -1, best_count = 0, niters = VH_RANSAC_ITERS, it_cur = 0;  // (thread 0's replay state, carried across the rounds)
    if (run) {
        for (int h0 = 0; h0 < s_bound; h0 += 8) {
            const int hyp = h0 + wave;
            if (hyp < VH_RANSAC_ITERS) {
                double M[6];
                int c = 0;
                if (ransac_hypothesis_lds(s_pairs, m, (uint32_t)hyp, M)) c = score_pairs(s_pairs, m, M, lane);
                if (lane == 0) s_counts[hyp] = c;
            }
            __syncthreads();
            if (tid == 0) {
                for (; it_cur < h0 + 8 && it_cur < niters; it_cur++) {
                    const int c = s_counts[it_cur];
                    if (c > max(best_count, 2)) {
                        r_best = it_cur;
                        best_count = c;
                        niters = ransac_update_iters(RANSAC_CONF, __ddiv_rn((double)(m - c), (double)m), niters);
                    }
                }
                s_bound = niters;
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        s_best = r_best;
        s_count = best_count;
        *J.bound = s_bound;
        if (r_best >= 0) {
            double M[6];
            ransac_hypothesis_lds(s_pairs, m, (uint32_t)r_best, M);
            for (int k = 0; k < 6; k++) s_M[k] = M[k];
        }
    }
    __syncthreads();
    const int best = s_best;
    if (best < 0) {
        for (int i = tid; i < n; i += 512) {
            J.inl[i] = 0;
            if (J.gate_valid) J.valid[i] = 0;
        }
        if (tid == 0) *J.status = 0;
        return;
    }
    double M[6];
    for (int k = 0; k < 6; k++) M[k] = s_M[k];
    auto block_sum = [&](long long* v, int nv) {  // exact int64 sums over the 8 wavefronts
        for (int k = 0; k < nv; k++) v[k] = vh_wave_sum_i64(v[k]);
        __syncthreads();
        if (lane == 0) for (int k = 0; k < nv; k++) s_red[k * 8 + wave] = v[k];
        __syncthreads();
        for (int k = 0; k < nv; k++) {
            long long t = 0;
            for (int w = 0; w < 8; w++) t += s_red[k * 8 + w];
            v[k] = t;
        }
    };
    // inlier mask over ALL points (0 where !valid) + the refit sums over the compacted pairs in LDS (the same points, the same integers)
    for (int i = tid; i < n; i += 512) J.inl[i] = 0;
    __syncthreads();
    long long s1[4] = {0, 0, 0, 0};
    unsigned in_bits = 0;  // this thread's pairs k = tid + 512 j, j < 6
#pragma unroll
    for (int j = 0; j < RANSAC_FUSED_MAX / 512; j++) {
        const int k = tid + 512 * j;
        if (k < m) {
            const float4 q = s_pairs[k];
            if (is_inlier(M, q.x, q.y, q.z, q.w)) {
                in_bits |= 1u << j;
                s1[0] += vh_fixq((double)q.x, 32); s1[1] += vh_fixq((double)q.y, 32);
                s1[2] += vh_fixq((double)q.z, 32); s1[3] += vh_fixq((double)q.w, 32);
                J.inl[s_idx[k]] = 1;
            }
        }
    }
    block_sum(s1, 4);
    const double cnt = (double)s_count;
    const double mx = __ddiv_rn(ldexp((double)s1[0], -32), cnt), my = __ddiv_rn(ldexp((double)s1[1], -32), cnt);
    const double mu = __ddiv_rn(ldexp((double)s1[2], -32), cnt), mv = __ddiv_rn(ldexp((double)s1[3], -32), cnt);
    long long q7[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < RANSAC_FUSED_MAX / 512; j++) {
        if ((in_bits >> j) & 1u) {
            const float4 q = s_pairs[tid + 512 * j];
            const double x = __dsub_rn((double)q.x, mx), y = __dsub_rn((double)q.y, my);
            const double u = __dsub_rn((double)q.z, mu), v = __dsub_rn((double)q.w, mv);
            q7[0] += vh_fixq(__dmul_rn(x, x), 20); q7[1] += vh_fixq(__dmul_rn(x, y), 20); q7[2] += vh_fixq(__dmul_rn(y, y), 20);
            q7[3] += vh_fixq(__dmul_rn(x, u), 20); q7[4] += vh_fixq(__dmul_rn(y, u), 20);
            q7[5] += vh_fixq(__dmul_rn(x, v), 20); q7[6] += vh_fixq(__dmul_rn(y, v), 20);
        }
    }
    block_sum(q7, 7);  // (its barriers also order the J.inl stores above before the reads below)
    if (J.gate_valid)
        for (int i = tid; i < n; i += 512) J.valid[i] = J.inl[i];
    if (tid == 0) {
        const double Sxx = ldexp((double)q7[0], -20), Sxy = ldexp((double)q7[1], -20), Syy = ldexp((double)q7[2], -20);
        const double Sxu = ldexp((double)q7[3], -20), Syu = ldexp((double)q7[4], -20);
        const double Sxv = ldexp((double)q7[5], -20), Syv = ldexp((double)q7[6], -20);
        const double det = __dsub_rn(__dmul_rn(Sxx, Syy), __dmul_rn(Sxy, Sxy));
        const double tr = __dadd_rn(Sxx, Syy);
        if (s_count >= 3 && det > __dmul_rn(__dmul_rn(1e-9, tr), tr) && det > 0) {
            const double a = __ddiv_rn(__dsub_rn(__dmul_rn(Sxu, Syy), __dmul_rn(Syu, Sxy)), det);
            const double b = __ddiv_rn(__dsub_rn(__dmul_rn(Sxx, Syu), __dmul_rn(Sxy, Sxu)), det);
            const double d = __ddiv_rn(__dsub_rn(__dmul_rn(Sxv, Syy), __dmul_rn(Syv, Sxy)), det);
            const double e = __ddiv_rn(__dsub_rn(__dmul_rn(Sxx, Syv), __dmul_rn(Sxy, Sxv)), det);
            M[0] = a; M[1] = b; M[2] = __dsub_rn(__dsub_rn(mu, __dmul_rn(a, mx)), __dmul_rn(b, my));
            M[3] = d; M[4] = e; M[5] = __dsub_rn(__dsub_rn(mv, __dmul_rn(d, mx)), __dmul_rn(e, my));
        }
        for (int k = 0; k < 6; k++) J.M[k] = M[k];
        *J.status = 1;
    }
}

// GLUE: the KLTmain glue that follows this RANSAC (vh_glue_dev.hpp) runs as the epilogue of the same workgroup -- 1: stage 1 -> 2 (mean translation, ROI,
// job descriptors, the rare shifted-crop copy), 2: stage 2 -> 3 (affine or fallback, remap job, fine-stage job); 0: stand-alone (vh_ransac_affine).  The
// glue reads what the body wrote to global memory (status, model, the gated validity flags): a workgroup-scope fence and a barrier order the two.
template <int GLUE>
__global__ __launch_bounds__(512) void k_ransac_fused(const void* tab, size_t stride, StreamWS* ws_all)
{
    ransac_fused_body(tab, stride);
    if constexpr (GLUE != 0) {
        __threadfence_block();  // (the same workgroup wrote and reads: one CU, one vector L1 -- as k_sess_frame orders its three parts)
        __syncthreads();
        if constexpr (GLUE == 1) klt_glue1_body<512>(ws_all[blockIdx.x]);
        else klt_glue2_body(ws_all[blockIdx.x]);
    }
}

static std::atomic<int> g_ransac_force{0};  // PROCESS-WIDE test hook (include/velocity_hip.h): 1 = always the three-kernel path, 2 = the fused kernel whenever the problem fits it
void vh_ransac_force_path(int mode) { g_ransac_force.store(mode, std::memory_order_relaxed); }

bool vh_launch_ransac(const void* job_tab, size_t tab_stride, int batch, int max_n, hipStream_t s, StreamWS* glue_ws, int glue)
{
    // up to RANSAC_FUSED_MAX pairs: one fused workgroup per stream, pairs / indices / counts resident in LDS (measured faster than the three
    // launches at every stream count, 1 .. 256: 4.53 -> 4.48 ms per step at 128 streams); more pairs: hypotheses spread over the chip
    const int force = g_ransac_force.load(std::memory_order_relaxed);
    if (max_n <= RANSAC_FUSED_MAX && (force == 2 || force == 0)) {
        const size_t lds = (size_t)RANSAC_FUSED_MAX * (16 + 4) + (size_t)VH_RANSAC_ITERS * 4;
        // 69 KB of dynamic LDS (> the 64 KB default limit): the attribute is per function AND per device; 0 = not tried, 1 = granted, -1 = refused
        // (three-kernel path)
        static std::atomic<signed char> attr_state[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev = dev < 0 || dev >= 64 ? 0 : dev;
        signed char st = attr_state[dev].load(std::memory_order_acquire);
        if (st == 0) {
            st = 1;
            for (const void* f : {reinterpret_cast<const void*>(k_ransac_fused<0>), reinterpret_cast<const void*>(k_ransac_fused<1>),
                                  reinterpret_cast<const void*>(k_ransac_fused<2>)})
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) st = -1;
            if (st < 0) (void)hipGetLastError();  // refused: forget the error, the three launches below serve every size
            attr_state[dev].store(st, std::memory_order_release);
        }
        if (st > 0) {
            const int g = glue_ws ? glue : 0;
            if (g == 1) hipLaunchKernelGGL(k_ransac_fused<1>, dim3(batch), dim3(512), lds, s, job_tab, tab_stride, glue_ws);
            else if (g == 2) hipLaunchKernelGGL(k_ransac_fused<2>, dim3(batch), dim3(512), lds, s, job_tab, tab_stride, glue_ws);
            else hipLaunchKernelGGL(k_ransac_fused<0>, dim3(batch), dim3(512), lds, s, job_tab, tab_stride, glue_ws);
            return g != 0;
        }
    }
    hipLaunchKernelGGL(k_ransac_compact, dim3(batch), dim3(1024), 0, s, job_tab, tab_stride);
    hipLaunchKernelGGL(k_ransac_score, dim3((VH_RANSAC_ITERS + 3) / 4, batch), dim3(256), 0, s, job_tab, tab_stride);
    hipLaunchKernelGGL(k_ransac_select, dim3(batch), dim3(256), 0, s, job_tab, tab_stride);
    return false;
}
