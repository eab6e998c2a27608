// Tracker session: the body of the reference's frame loop (vidExample.py:133-160) for `batch` independent video
// streams, entirely on the device -- KLTmain, the track-state bookkeeping (K19: vg[vg]=v, vp&=vg, stream compaction,
// pose-track selection, NaN-padded history P), estimateWorldCameraPose(findR=False), the B/S result records and, at
// the MSV frame, fcnMSV1_t.  One step = a fixed sequence of launches over all streams, no host round trip.
#include <math.h>
#include <string.h>

#include <new>

#include <stddef.h>

#include "vh_ws.hpp"
#include "vh_pose_dev.hpp"

// History P (vidExample.py:128-129: [5, N0, n] in the reference) is kept FRAME-MAJOR on the device, [nhist][5][N0]: a frame writes its five rows as five
// contiguous runs.  In the reference's layout the entries of one frame are nhist floats apart: 10 000 single-float stores per stream and frame, each into
// its own cache line -- at 256 streams 2.6 M partial-line writes per step, which is what made k_sess_frame take 140 us there and 70 us for one stream.
// velocity_amd/driver.py::TrackerSession.state() hands the reference's [5, N0, nhist] view to the caller.
__device__ __host__ __forceinline__ size_t sess_P(int row, int track, int frame, int N0) { return ((size_t)frame * 5 + row) * (size_t)N0 + track; }

struct vh_session {
    vh_ctx* ctx;
    int batch, N0, nhist, w, h, msv_frame, k_is_f32;
    vh_lk_params coarse, fine;
    char* arena;
    SessStream* d_ss;
    SessStream* h_ss;  // host mirror of the pointer fields
    IngestJob* d_ingest;  // [batch] descriptors of the fused BGR ingest
    int* h_frame;      // per stream: frames stepped since its vh_session_init (host mirror of SessStream::frame_i; the
                       // reference fires fcnMSV1_t at `i == msvFrame` of EACH video, vidExample.py:155)
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
#define SESS_CHECK() VH_CHECK(hipGetLastError())

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
// vg[vg] = v ; vp &= vg ; p = p[v] ; pose selection p[vp[vg]] / p3[vp]   (vidExample.py:134-139)
// A thread owns up to 16 CONSECUTIVE current tracks: it loads all of them at once (one memory round trip, not one per 256-track chunk), both
// order-preserving compactions -- the surviving tracks and, among them, the pose tracks -- come from one block-wide exclusive scan of the two
// per-thread counts.  `vp &= vg` only ever changes the entries of CURRENT tracks (a dropped track's vg and vp are already 0), so the two mask
// updates are per-track stores instead of passes over all N0 entries.
template <int NT>
__device__ __forceinline__ void sess_book_a(SessStream& S)
{
    constexpr int EPT = 4096 / NT, NW = NT / 64;
    const int tid = threadIdx.x, n = S.n_cur, wave = tid >> 6, lane = tid & 63;
    __shared__ int s_tot[2][NW], s_base[2];
    uint8_t* const vg = S.vg;
    uint8_t* const vp = S.vp;
    int* const ids = S.ids;
    if (tid == 0) { s_base[0] = 0; s_base[1] = 0; }
    __syncthreads();
    for (int c0 = 0; c0 < n || c0 == 0; c0 += NT * EPT) {  // one pass for up to 4096 tracks
        const int per = min(EPT, (min(n - c0, NT * EPT) + NT - 1) / NT);
        const int b = c0 + tid * per, e = min(n, b + per);
        int id[EPT];
        float px[EPT], py[EPT];
        unsigned fv = 0, fp = 0;  // bit k: track b + k survives / is a pose track
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int i = b + k;
            if (k < per && i < e) {
                id[k] = ids[i];
                px[k] = S.p_all[2 * i]; py[k] = S.p_all[2 * i + 1];
                fv |= (S.v[i] != 0 ? 1u : 0u) << k;
            }
        }
#pragma unroll
        for (int k = 0; k < EPT; k++)
            if (k < per && b + k < e) {
                const bool f = (fv >> k) & 1u;
                const bool pv = f && vp[id[k]] != 0;
                vg[id[k]] = f ? 1 : 0;
                vp[id[k]] = pv ? 1 : 0;
                fp |= (pv ? 1u : 0u) << k;
            }
        const int cv = __popc(fv), cp = __popc(fp);
        // block-wide exclusive scan of (cv, cp): wave scan by shuffles, wave totals through LDS
        int sv = cv, sp = cp;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int tv = __shfl_up(sv, o, 64), tp = __shfl_up(sp, o, 64);
            if (lane >= o) { sv += tv; sp += tp; }
        }
        if (lane == 63) { s_tot[0][wave] = sv; s_tot[1][wave] = sp; }
        __syncthreads();  // also: every thread has LOADED its tracks before anybody overwrites ids / p_cur below
        int ov = s_base[0] + sv - cv, op = s_base[1] + sp - cp;
        for (int w = 0; w < wave; w++) { ov += s_tot[0][w]; op += s_tot[1][w]; }
#pragma unroll
        for (int k = 0; k < EPT; k++)
            if (k < per && ((fv >> k) & 1u)) {
                ids[ov] = id[k];
                S.p_cur[2 * ov] = px[k]; S.p_cur[2 * ov + 1] = py[k];
                if ((fp >> k) & 1u) { S.sel_p[op] = ov; S.sel_pw[op] = id[k]; op++; }
                ov++;
            }
        __syncthreads();
        if (tid == 0) {
            for (int w = 0; w < NW; w++) { s_base[0] += s_tot[0][w]; s_base[1] += s_tot[1][w]; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        S.n_cur = s_base[0];
        S.n_pose = s_base[1];
    }
}

// results records B, S and history P (vidExample.py:142-146, 151-153, 164); then advance the frame state
template <int NT>
__device__ __forceinline__ void sess_book_b(SessStream& S, const uint8_t* const* frames, float time_s, float frame_no, const float* times,
                                            const float* frame_nos)
{
    if (times) time_s = times[blockIdx.x];          // independent videos: per-stream timestamp (B[i,12] = CAP_PROP_POS_MSEC / 1000)
    if (frame_nos) frame_no = frame_nos[blockIdx.x];
    const int tid = threadIdx.x, i = S.frame_i + 1, nh = S.nhist, N0 = S.N0;
    if (i < nh) {
        for (int k = tid; k < S.n_cur; k += NT) {
            const int g = S.ids[k];
            S.P[sess_P(0, g, i, N0)] = S.p_cur[2 * k];
            S.P[sess_P(1, g, i, N0)] = S.p_cur[2 * k + 1];
            S.P[sess_P(4, g, i, N0)] = (float)i;
        }
        for (int j = tid; j < S.n_pose; j += NT) {
            const int g = S.sel_pw[j];
            S.P[sess_P(2, g, i, N0)] = (float)S.p_proj[2 * j];
            S.P[sess_P(3, g, i, N0)] = (float)S.p_proj[2 * j + 1];
        }
    }
    if (tid == 0) {
        if (i < nh) {
            float* B = S.B;
            B[14 * i + 12] = time_s;
            B[14 * i + 13] = frame_no;
            const float dt = __fsub_rn(B[14 * i + 12], B[14 * (i - 1) + 12]);
            // dr = norm(t + B[0,0:3] - B[i-1,0:3]) in float32 (vidExample.py:143)
            float ss = 0.f;
            for (int c = 0; c < 3; c++) {
                const float d = __fsub_rn(__fadd_rn(S.t[c], B[c]), B[14 * (i - 1) + c]);
                ss = __fadd_rn(ss, __fmul_rn(d, d));
            }
            const float dr = vh_sqrtf(ss);
            S.r_total = __fadd_rn(S.r_total, dr);
            for (int c = 0; c < 3; c++) {
                B[14 * i + 3 + c] = S.t[c];
                B[14 * i + c] = __fadd_rn(B[c], S.t[c]);
            }
            float* R = S.S + 9 * i;
            R[0] = (float)i; R[1] = 0.f; R[2] = (float)S.n_cur; R[3] = (float)S.res; R[4] = dt;
            R[5] = __fsub_rn(B[14 * i + 12], S.t0); R[6] = dr; R[7] = S.r_total;
            R[8] = __fmul_rn(__fdiv_rn(dr, dt), 3.6f);
        }
        S.im0 = frames[blockIdx.x];
        S.pp ^= 1;
        S.frame_i = i;
    }
}

__global__ __launch_bounds__(256) void k_sess_book_a(SessStream* ss_all) { sess_book_a<256>(ss_all[blockIdx.x]); }
__global__ __launch_bounds__(256) void k_sess_book_b(SessStream* ss_all, const uint8_t* const* frames, float time_s, float frame_no,
                                                     const float* times, const float* frame_nos)
{
    sess_book_b<256>(ss_all[blockIdx.x], frames, time_s, frame_no, times, frame_nos);
}

// The whole post-tracking part of a frame in ONE launch (N0 <= 4096): vg[vg] = v, vp &= vg, compaction and pose-track selection
// (vidExample.py:135-139), estimateWorldCameraPose(findR=False) with every LM iteration (NLS.py:9-33,102-129), then the B / S / P records
// (vidExample.py:142-153,164).  Three dependent launches of one workgroup each before; same code, same results.
// 512 threads: the two bookkeeping halves are chains of dependent memory round trips and the LM loop a chain of dependent float64 operations, so two
// wavefronts per SIMD hide what one cannot: 67 -> 56 us at 256 streams (the stand-alone pose kernel gains nothing from it: 2.75 vs 3.05 us per iteration)
#define SESS_NT 512
__global__ __launch_bounds__(SESS_NT) void k_sess_frame(SessStream* ss_all, const uint8_t* const* frames, float time_s, float frame_no,
                                                        const float* times, const float* frame_nos)
{
    SessStream& S = ss_all[blockIdx.x];
    sess_book_a<SESS_NT>(S);
    __threadfence_block();
    __syncthreads();
    const PoseJob J = S.pose;
    pose_solve<0, SESS_NT>(J);
    __threadfence_block();
    __syncthreads();
    sess_book_b<SESS_NT>(S, frames, time_s, frame_no, times, frame_nos);
}

// p3[vg] = p3hat - t ; vp = vg   (vidExample.py:159-160)
__global__ __launch_bounds__(256) void k_sess_after_msv(SessStream* ss_all, int msv_frame)
{
    SessStream& S = ss_all[blockIdx.x];
    if (S.frame_i != msv_frame) return;  // only the streams whose own frame counter is at the MSV frame
    for (int k = threadIdx.x; k < S.n_cur; k += 256) {
        const int g = S.ids[k];
        for (int c = 0; c < 3; c++) S.p3[3 * g + c] = S.msv_b0[3 * k + c] - (double)S.t[c];
    }
    for (int g = threadIdx.x; g < S.N0; g += 256) S.vp[g] = S.vg[g];
}

// frame-0 initialisation (vidExample.py:125-131, 151-153)
// t0_dev / res0_dev / n_dev (all may be null): plate pose, its residual and the number of frame-0 tracks taken from DEVICE memory -- the outputs of
// vh_frame0_init -- instead of the by-value arguments; with n_dev < N0 the tail rows are tracks that never existed (vg 0, history NaN)
__global__ __launch_bounds__(256) void k_sess_init(SessStream* ss, const float* p, const double* p3, const uint8_t* vp, const uint8_t* frame0,
                                                   float t0x, float t0y, float t0z, float time0, float frame_no, float res0, const float* t0_dev,
                                                   const double* res0_dev, const int* n_dev)
{
    SessStream& S = *ss;
    const int tid = threadIdx.x, N0 = S.N0, nh = S.nhist;
    const int n = n_dev ? min(max(*n_dev, 0), N0) : N0;
    if (t0_dev) { t0x = t0_dev[0]; t0y = t0_dev[1]; t0z = t0_dev[2]; }
    if (res0_dev) res0 = (float)*res0_dev;
    const float nanv = __int_as_float(0x7fc00000);
    for (size_t q = tid; q < (size_t)5 * N0 * nh; q += 256) S.P[q] = nanv;
    for (int q = tid; q < nh * 14; q += 256) S.B[q] = 0.f;
    for (int q = tid; q < nh * 9; q += 256) S.S[q] = 0.f;
    __syncthreads();
    for (int g = n + tid; g < N0; g += 256) { S.vg[g] = 0; S.vp[g] = 0; }
    for (int g = tid; g < n; g += 256) {
        S.vg[g] = 1;
        S.vp[g] = vp[g] ? 1 : 0;
        S.ids[g] = g;
        S.p_cur[2 * g] = p[2 * g]; S.p_cur[2 * g + 1] = p[2 * g + 1];
        for (int c = 0; c < 3; c++) S.p3[3 * g + c] = p3[3 * g + c];
        S.P[sess_P(0, g, 0, N0)] = p[2 * g];
        S.P[sess_P(1, g, 0, N0)] = p[2 * g + 1];
        if (vp[g]) { S.P[sess_P(2, g, 0, N0)] = p[2 * g]; S.P[sess_P(3, g, 0, N0)] = p[2 * g + 1]; }  // p_ = p[vp]
        S.P[sess_P(4, g, 0, N0)] = 0.f;
    }
    if (tid == 0) {
        S.n_cur = n; S.n_pose = 0; S.frame_i = 0; S.pp = 0; S.klt_flags = 0; S.small_ready = 0;
        S.B[0] = t0x; S.B[1] = t0y; S.B[2] = t0z; S.B[12] = time0; S.B[13] = frame_no;
        S.t[0] = t0x; S.t[1] = t0y; S.t[2] = t0z;
        S.t0 = time0; S.r_total = 0.f; S.res = res0_dev ? *res0_dev : (double)res0;
        S.S[0] = 0.f; S.S[2] = (float)n; S.S[3] = res0; S.S[4] = nanv; S.S[8] = nanv;
        S.im0 = frame0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
// fcnMSV1_t (vidExample.py:155-160) can fire for this session: the predicate that both sizes its scratch and gates its launch
static inline bool sess_msv_fires(int msv_frame, int nhist) { return msv_frame >= 1 && msv_frame + 1 <= 2048 && msv_frame < nhist; }

extern "C" VH_API int vh_session_create(vh_session** out, vh_ctx* ctx, int n0, int nhist, int w, int h, const double* K_host,
                                        int k_is_float32, const vh_lk_params* coarse, const vh_lk_params* fine, int msv_frame)
{
    if (!out || !ctx || !K_host || !coarse || !fine || n0 < 1 || nhist < 2) return vh_fail(-1, "vh_session_create: bad arguments");
    if (n0 > ctx->max_pts || w > ctx->max_w || h > ctx->max_h) return vh_fail(-1, "vh_session_create: exceeds the workspace");
    // a re-triangulation frame the session COULD reach (< nhist) but fcnMSV1_t cannot serve is an error, not a silently skipped step
    if (msv_frame >= 1 && msv_frame < nhist && msv_frame + 1 > 2048)
        return vh_fail(-1, "vh_session_create: msv_frame + 1 > 2048 frames (fcnMSV1_t keeps one ray table entry per frame of the history in LDS; the "
                           "reference has no such limit -- pass msv_frame <= 0 to run without the re-triangulation, or a frame below 2048)");
    vh_session* s = new (std::nothrow) vh_session();
    if (!s) return vh_fail(-1, "out of host memory");
    s->ctx = ctx; s->batch = ctx->batch; s->N0 = n0; s->nhist = nhist; s->w = w; s->h = h; s->msv_frame = msv_frame; s->k_is_f32 = k_is_float32 ? 1 : 0;
    s->coarse = *coarse; s->fine = *fine;
    const int dw = (int)lrint(w * 0.25), dh = (int)lrint(h * 0.25);
    s->h_ss = new SessStream[s->batch];
    memset(s->h_ss, 0, sizeof(SessStream) * s->batch);
    s->h_frame = new int[s->batch]();
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            hipError_t e = hipMalloc((void**)&s->arena, off);
            if (e != hipSuccess) { delete[] s->h_ss; delete[] s->h_frame; delete s; vh_set_error("hipMalloc(session)", e, __FILE__, __LINE__); return (int)e; }
            (void)hipMemset(s->arena, 0, off);
            off = 0;
        }
        char* base = pass ? s->arena : nullptr;
        s->d_ss = (SessStream*)(base + carve(sizeof(SessStream) * s->batch));
        s->d_ingest = (IngestJob*)(base + carve(sizeof(IngestJob) * s->batch));
        for (int b = 0; b < s->batch; b++) {
            SessStream& S = s->h_ss[b];
            S.vg = (uint8_t*)(base + carve(n0)); S.vp = (uint8_t*)(base + carve(n0));
            S.p_cur = (float*)(base + carve(sizeof(float) * 2 * n0)); S.ids = (int*)(base + carve(sizeof(int) * n0));
            S.p3 = (double*)(base + carve(sizeof(double) * 3 * n0));
            S.P = (float*)(base + carve(sizeof(float) * 5 * (size_t)n0 * nhist));
            S.B = (float*)(base + carve(sizeof(float) * 14 * nhist)); S.S = (float*)(base + carve(sizeof(float) * 9 * nhist));
            S.p_all = (float*)(base + carve(sizeof(float) * 2 * n0)); S.v = (uint8_t*)(base + carve(n0));
            S.sel_p = (int*)(base + carve(sizeof(int) * n0)); S.sel_pw = (int*)(base + carve(sizeof(int) * n0));
            S.p_proj = (double*)(base + carve(sizeof(double) * 2 * n0));
            S.msv_U = (double*)(base + carve(sizeof(double) * 3 * (size_t)(sess_msv_fires(msv_frame, nhist) ? msv_frame + 1 : 1) * n0));  /* one row when fcnMSV1_t can never fire (a 'never' sentinel must not size the buffer) */ S.msv_b0 = (double*)(base + carve(sizeof(double) * 3 * n0));
            S.small[0] = (uint8_t*)(base + carve((size_t)dw * dh)); S.small[1] = (uint8_t*)(base + carve((size_t)dw * dh));
        }
    }
    for (int b = 0; b < s->batch; b++) {
        SessStream& S = s->h_ss[b];
        SessStream* d = s->d_ss + b;
        for (int k = 0; k < 9; k++) S.K[k] = K_host[k];
        S.N0 = n0; S.nhist = nhist; S.w = w; S.h = h; S.stride = w;
        PoseJob& J = S.pose;
        for (int k = 0; k < 9; k++) { J.K[k] = S.K[k]; J.R[k] = (k % 4 == 0) ? 1.0 : 0.0; }
        J.x0[0] = J.x0[1] = J.x0[2] = 0; J.x0[3] = 0; J.x0[4] = 0; J.x0[5] = 1;  // always restarted from t=[0,0,1] (NLS.py:9)
        J.p = S.p_cur; J.pw = S.p3; J.p_sel = S.sel_p; J.pw_sel = S.sel_pw; J.n_ptr = &d->n_pose; J.n = 0; J.mode = 0;
        J.t_out = d->t; J.R_out = nullptr; J.res_out = &d->res; J.p_proj = S.p_proj; J.info_out = d->pose_info;
        MsvJob& M = S.msv;
        memset(&M, 0, sizeof(M));
        for (int k = 0; k < 9; k++) M.K[k] = S.K[k];
        M.P = S.P; M.P_rs = (size_t)n0; M.P_ts = 1; M.P_fs = (size_t)5 * n0;  // the session's frame-major history
        M.B = S.B; M.ids = S.ids; M.ng_ptr = &d->n_cur; M.ng = 0; M.N0 = n0; M.nhist = nhist;
        M.nf = msv_frame + 1; M.max_iter = 1000; M.f32_rays = s->k_is_f32;  // P is float32 always; K as the caller holds it
        M.U = S.msv_U; M.b0 = S.msv_b0; M.x_out = d->msv_x; M.info_out = d->msv_info;
    }
    hipError_t e = hipMemcpy(s->d_ss, s->h_ss, sizeof(SessStream) * s->batch, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(s->arena); delete[] s->h_ss; delete[] s->h_frame; delete s; vh_set_error("hipMemcpy(session)", e, __FILE__, __LINE__); return (int)e; }
    *out = s;
    return 0;
}

extern "C" VH_API void vh_session_destroy(vh_session* s)
{
    if (!s) return;
    (void)hipFree(s->arena);
    delete[] s->h_ss;
    delete[] s->h_frame;
    delete s;
}

static int session_init(vh_session* s, int slot, const uint8_t* frame0, int stride, const float* p, const double* p3, const uint8_t* vp,
                        const float* t0_host, float time0, float frame_no, float res0, const float* t0_dev, const double* res0_dev, const int* n_dev,
                        void* stream)
{
    if (stride != s->w) return vh_fail(-1, "vh_session_init: frames must be dense (stride == width)");
    VH_BIND(s->ctx, stream);
    hipStream_t st = bound_.s;
    hipLaunchKernelGGL(k_sess_init, dim3(1), dim3(256), 0, st, s->d_ss + slot, p, p3, vp, frame0, t0_host ? t0_host[0] : 0.f, t0_host ? t0_host[1] : 0.f,
                       t0_host ? t0_host[2] : 0.f, time0, frame_no, res0, t0_dev, res0_dev, n_dev);
    // quarter-scale copy of frame 0 = im0_small of the first step (pp starts at 0 -> previous index 1)
    int r = vh_resize_quarter(s->ctx, frame0, s->w, s->h, stride, s->h_ss[slot].small[1], stream);
    if (r) return r;
    s->h_frame[slot] = 0;  // a (re-)initialised slot starts a new clip: its MSV frame counts from here
    SESS_CHECK();
    return 0;
}

extern "C" VH_API int vh_session_init(vh_session* s, int slot, const uint8_t* frame0, int stride, const float* p, const double* p3,
                                      const uint8_t* vp, const float* t0_host, float time0, float frame_no, float res0, void* stream)
{
    if (!s || slot < 0 || slot >= s->batch || !t0_host) return vh_fail(-1, "vh_session_init: bad arguments");
    return session_init(s, slot, frame0, stride, p, p3, vp, t0_host, time0, frame_no, res0, nullptr, nullptr, nullptr, stream);
}

extern "C" VH_API int vh_session_init_dev(vh_session* s, int slot, const uint8_t* frame0, int stride, const float* p, const double* p3,
                                          const uint8_t* vp, const float* t0_dev, const double* res0_dev, const int* n_dev, float time0,
                                          float frame_no, void* stream)
{
    if (!s || slot < 0 || slot >= s->batch || !t0_dev || !res0_dev) return vh_fail(-1, "vh_session_init_dev: bad arguments");
    return session_init(s, slot, frame0, stride, p, p3, vp, nullptr, time0, frame_no, 0.f, t0_dev, res0_dev, n_dev, stream);
}

static int session_step(vh_session* s, const uint8_t* const* frames_dev, float time_s, float frame_no, const float* times_dev,
                        const float* frame_nos_dev, void* stream)
{
    if (!s || !frames_dev) return vh_fail(-1, "vh_session_step: bad arguments");
    VH_BIND(s->ctx, stream);
    hipStream_t st = bound_.s;
    vh_ctx* c = s->ctx;
    const int nb = s->batch;
    int r = vh_run_klt_main(c, 0, nb, st, s->coarse, s->fine, s->d_ss, frames_dev, s->N0);  // the set-up kernel also fetches this frame's KltIO from the session
    if (r) return r;
    if (s->N0 <= 4096) {
        const int rec = vh_prof_start(s->ctx, st);
        hipLaunchKernelGGL(k_sess_frame, dim3(nb), dim3(SESS_NT), 0, st, s->d_ss, frames_dev, time_s, frame_no, times_dev, frame_nos_dev);
        vh_prof_stop(s->ctx, rec, VH_PROF_SESSION, st);
    } else {  // more pose tracks than 256 threads keep in registers: the 1024-thread pose kernel between the two bookkeeping halves
        hipLaunchKernelGGL(k_sess_book_a, dim3(nb), dim3(256), 0, st, s->d_ss);
        vh_launch_pose(&s->d_ss[0].pose, sizeof(SessStream), nb, 0, s->N0, st);
        hipLaunchKernelGGL(k_sess_book_b, dim3(nb), dim3(256), 0, st, s->d_ss, frames_dev, time_s, frame_no, times_dev, frame_nos_dev);
    }
    // fcnMSV1_t fires when a stream reaches ITS frame msv_frame (vidExample.py:155), whenever that stream was initialised
    const bool msv_ok = sess_msv_fires(s->msv_frame, s->nhist);
    bool any = false;
    for (int b = 0; b < nb; b++) {
        const int fi = ++s->h_frame[b];
        if (msv_ok && fi == s->msv_frame) any = true;
    }
    // ONE launch for every stream: a workgroup runs only when its stream's own frame counter is at the MSV frame (k_msv1_tab)
    if (any) vh_launch_msv1_tab(&s->d_ss[0].msv, sizeof(SessStream), (ptrdiff_t)offsetof(SessStream, frame_i) - (ptrdiff_t)offsetof(SessStream, msv),
                                s->msv_frame, s->msv_frame + 1, nb, st);
    if (any) hipLaunchKernelGGL(k_sess_after_msv, dim3(nb), dim3(256), 0, st, s->d_ss, s->msv_frame);
    SESS_CHECK();
    return 0;
}

extern "C" VH_API int vh_session_step(vh_session* s, const uint8_t* const* frames_dev, float time_s, float frame_no, void* stream)
{
    return session_step(s, frames_dev, time_s, frame_no, nullptr, nullptr, stream);
}

extern "C" VH_API int vh_session_step_v(vh_session* s, const uint8_t* const* frames_dev, const float* time_s_dev, const float* frame_no_dev,
                                        void* stream)
{
    if (!time_s_dev || !frame_no_dev) return vh_fail(-1, "vh_session_step_v: bad arguments");
    return session_step(s, frames_dev, 0.f, 0.f, time_s_dev, frame_no_dev, stream);
}

// packed track state of every stream for the cross-GPU exchange (K20): per stream a record of 8 + 3*N0 float32 words
// [n_cur, n_pose, frame_i, klt_flags, t0, t1, t2, res | p (N0 x 2, zero padded) | ids (N0 int32 bit patterns, -1 padded)]
__global__ __launch_bounds__(256) void k_sess_pack(const SessStream* ss_all, float* out, int rec)
{
    const SessStream& S = ss_all[blockIdx.x];
    float* o = out + (size_t)blockIdx.x * rec;
    const int N0 = S.N0, n = S.n_cur;
    if (threadIdx.x == 0) {
        o[0] = (float)n; o[1] = (float)S.n_pose; o[2] = (float)S.frame_i; o[3] = (float)S.klt_flags;
        o[4] = S.t[0]; o[5] = S.t[1]; o[6] = S.t[2]; o[7] = (float)S.res;
    }
    for (int k = threadIdx.x; k < N0; k += 256) {
        o[8 + 2 * k] = k < n ? S.p_cur[2 * k] : 0.f;
        o[8 + 2 * k + 1] = k < n ? S.p_cur[2 * k + 1] : 0.f;
        o[8 + 2 * N0 + k] = __int_as_float(k < n ? S.ids[k] : -1);
    }
}

// fused frame ingest for every stream: descriptors from the stream state (the quarter-scale image goes straight into
// the buffer the coming step reads as im_small), then ONE pass over the BGR frames
__global__ void k_sess_ingest_jobs(SessStream* ss_all, IngestJob* jobs, const uint8_t* const* bgr, int bgr_stride, uint8_t* const* gray, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    SessStream& S = ss_all[b];
    IngestJob J;
    J.bgr = bgr[b]; J.gray = gray[b]; J.small = S.small[S.pp];
    J.w = S.w; J.h = S.h; J.bgr_stride = bgr_stride; J.gray_stride = S.stride;
    J.dw = __double2int_rn(S.w * 0.25); J.dh = __double2int_rn(S.h * 0.25); J.small_stride = J.dw;
    jobs[b] = J;
    if (J.bgr != nullptr) S.small_ready = S.frame_i + 1;
}

extern "C" VH_API int vh_session_ingest_bgr(vh_session* s, const uint8_t* const* bgr_frames_dev, int bgr_stride, uint8_t* const* gray_frames_dev, void* stream)
{
    if (!s || !bgr_frames_dev || !gray_frames_dev || bgr_stride < 3 * s->w) return vh_fail(-1, "vh_session_ingest_bgr: bad arguments");
    VH_BIND(s->ctx, stream);
    hipStream_t st = bound_.s;
    hipLaunchKernelGGL(k_sess_ingest_jobs, dim3((s->batch + 63) / 64), dim3(64), 0, st, s->d_ss, s->d_ingest, bgr_frames_dev, bgr_stride, gray_frames_dev, s->batch);
    vh_launch_ingest_bgr(s->d_ingest, s->batch, s->w, s->h, st);
    SESS_CHECK();
    return 0;
}

extern "C" VH_API int vh_session_pack_state(vh_session* s, float* out, void* stream)
{
    if (!s || !out) return vh_fail(-1, "vh_session_pack_state: bad arguments");
    hipLaunchKernelGGL(k_sess_pack, dim3(s->batch), dim3(256), 0, (hipStream_t)stream, s->d_ss, out, 8 + 3 * s->N0);
    SESS_CHECK();
    return 0;
}

extern "C" VH_API int vh_session_ptrs(vh_session* s, int slot, vh_session_view* out)
{
    if (!s || !out || slot < 0 || slot >= s->batch) return vh_fail(-1, "vh_session_ptrs: bad arguments");
    const SessStream& H = s->h_ss[slot];
    SessStream* d = s->d_ss + slot;
    out->vg = H.vg; out->vp = H.vp; out->p = H.p_cur; out->ids = H.ids; out->p3 = H.p3; out->P = H.P; out->B = H.B; out->S = H.S;
    out->n_cur = &d->n_cur; out->n_pose = &d->n_pose; out->t = d->t; out->res = &d->res; out->frame_i = &d->frame_i;
    out->klt_flags = &d->klt_flags; out->pose_info = d->pose_info; out->sel_pw = H.sel_pw; out->p_proj = H.p_proj;
    // history layout, in floats: entry (row, track, frame) of P at P[row * P_row_stride + track * P_track_stride + frame * P_frame_stride]
    out->P_row_stride = (size_t)H.N0; out->P_track_stride = 1; out->P_frame_stride = (size_t)5 * H.N0;
    out->n0 = H.N0; out->nhist = H.nhist;
    return 0;
}
