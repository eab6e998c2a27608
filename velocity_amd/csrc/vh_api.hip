// C-ABI entry points (include/velocity_hip.h), the device workspace, and the KLTmain stage pipeline
// (utils/KLT.py:99-134).  Data-dependent quantities (track count, ROI, shift, transforms, pyramid level sizes) never
// leave the GPU: small "glue" kernels write the job descriptors of the next stage into the per-stream workspace, and
// every image / track kernel is launched over the maximum extent and reads its extent from there.  A frame is thus a
// fixed sequence of launches with no host round trip (hipGraph-capturable), for any number of streams per launch.
#include <atomic>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "vh_ba.hpp"
#include "vh_ws.hpp"
#include "vh_glue_dev.hpp"

// ---------------------------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vh_set_error(const char* what, hipError_t e, const char* file, int line)
{
    snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}
int vh_fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
#define VH_LAUNCH_CHECK() VH_CHECK(hipGetLastError())

extern "C" VH_API int vh_version(void) { return 105; }
void vh_lk_force_generic(int on);
extern "C" VH_API void vh_debug_force_generic_lk(int on) { vh_lk_force_generic(on); }
void vh_ransac_force_path(int mode);
extern "C" VH_API void vh_debug_ransac_path(int mode) { vh_ransac_force_path(mode); }

extern "C" VH_API int vh_copy_to_host(void* dst_host, const void* src_dev, size_t bytes, void* stream)
{
    VH_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    VH_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
extern "C" VH_API const char* vh_last_error(void) { return g_err; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" VH_API int vh_ctx_create(vh_ctx** out, int batch, int max_w, int max_h, int max_pts)
{
    if (!out || batch < 1 || max_w < 4 || max_h < 4 || max_pts < 1) return vh_fail(-1, "vh_ctx_create: bad arguments");
    vh_ctx* c = new (std::nothrow) vh_ctx();
    if (!c) return vh_fail(-1, "vh_ctx_create: out of host memory");
    c->batch = batch; c->max_w = max_w; c->max_h = max_h; c->max_pts = max_pts;
    c->sw = (int)lrint(max_w * 0.25); c->sh = (int)lrint(max_h * 0.25);
    // carve plan (computed twice: size, then pointers)
    StreamBufs* hb = new StreamBufs[batch];
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // a level buffer holds either layout of any level up to w x h: dense, or bordered (small levels, see VH_LV_PAD)
    auto lv_bytes = [](int w, int h) { return (size_t)VH_LV_STRIDE(w) * (size_t)(h + 2 * VH_LV_PAD); };
    size_t ws_off = carve(sizeof(StreamWS) * batch);
    size_t small_off = carve(sizeof(double) * 64);
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            c->arena_bytes = off;
            hipError_t e = hipMalloc((void**)&c->arena, off);
            if (e != hipSuccess) { delete[] hb; delete c; vh_set_error("hipMalloc(arena)", e, __FILE__, __LINE__); return (int)e; }
            e = hipMemset(c->arena, 0, off);
            if (e != hipSuccess) { (void)hipFree(c->arena); delete[] hb; delete c; vh_set_error("hipMemset(arena)", e, __FILE__, __LINE__); return (int)e; }
            off = 0;
            carve(sizeof(StreamWS) * batch);
            carve(sizeof(double) * 64);
        }
        char* base = pass ? c->arena : nullptr;
        for (int b = 0; b < batch; b++) {
            StreamBufs& B = hb[b];
            for (int k = 0; k < 2; k++) {
                B.small0[k] = (uint8_t*)(base + carve((size_t)c->sw * c->sh));
                int w = c->sw, h = c->sh;
                for (int l = 1; l < VH_MAX_LEVELS; l++) {
                    w = (w + 1) / 2; h = (h + 1) / 2;
                    B.small_lv[k][l] = (uint8_t*)(base + carve(lv_bytes(w, h)));
                }
                B.small_lv[k][0] = nullptr;
                w = max_w; h = max_h;
                for (int l = 1; l < VH_MAX_LEVELS; l++) {
                    w = (w + 1) / 2; h = (h + 1) / 2;
                    B.roi_lv[k][l] = (uint8_t*)(base + carve(lv_bytes(w, h)));
                }
                B.roi_lv[k][0] = nullptr;
            }
            B.warp = (uint8_t*)(base + carve((size_t)((max_w + 3) & ~3) * max_h));
            B.p_small = (float*)(base + carve(sizeof(float) * 2 * max_pts));
            B.p_coarse = (float*)(base + carve(sizeof(float) * 2 * max_pts));
            B.v_small = (uint8_t*)(base + carve(max_pts));
            B.v_coarse = (uint8_t*)(base + carve(max_pts));
            B.v_all = (uint8_t*)(base + carve(max_pts));
            B.inl = (uint8_t*)(base + carve(max_pts));
            B.idx = (int*)(base + carve(sizeof(int) * max_pts));
            B.pairs = (float4*)(base + carve(sizeof(float4) * max_pts));
            B.counts = (int*)(base + carve(sizeof(int) * VH_RANSAC_ITERS));
            B.order = (int*)(base + carve(sizeof(int) * max_pts));
        }
    }
    c->d_ws = (StreamWS*)(c->arena + ws_off);
    c->d_small = (double*)(c->arena + small_off);
    hipError_t e = hipSuccess;
    for (int b = 0; b < batch && e == hipSuccess; b++) {
        e = hipMemcpy(&c->d_ws[b].bufs, &hb[b], sizeof(StreamBufs), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(hb[b].v_all, 1, max_pts);
    }
    if (e != hipSuccess) { (void)hipFree(c->arena); delete[] hb; delete c; vh_set_error("hipMemcpy(bufs)", e, __FILE__, __LINE__); return (int)e; }
    c->h_bufs = hb;
    *out = c;
    return 0;
}

extern "C" VH_API void vh_ctx_destroy(vh_ctx* c)
{
    if (!c) return;
    vh_ba_graph_cache_free(c->ba_graphs);
    vh_init_scratch_free(c);
    (void)hipFree(c->arena);
    if (c->bound_ev) (void)hipEventDestroy(c->bound_ev);
    for (int k = 0; k < 2 * c->prof_cap; k++) (void)hipEventDestroy(c->prof_ev[k]);
    delete[] c->prof_ev;
    delete[] c->prof_stage;
    delete[] c->h_bufs;
    delete c;
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers shared by the glue kernels
// ---------------------------------------------------------------------------------------------------------------
// ---- stage 0: descriptors of the quarter-scale stage (KLT.py:110-114) -------------------------------------------
__device__ void klt_setup_descriptors(StreamWS& ws, const SessStream* ss_all, const uint8_t* const* frames, const vh_lk_params& coarse,
                                      const vh_lk_params& fine, int use_order);

// Launch order of a stream's tracks (use_order): the LK kernels read every track's windows out of the pyramids, and neighbouring workgroups run at the same
// time on the same XCD (lk_block_xy) -- tracks that are neighbours in the IMAGE should be neighbours in the LAUNCH.  The reference's tracks come from
// goodFeaturesToTrack, sorted by corner response, i.e. spatially at random: at 256 streams the ROI-stage launch then takes 1465 us instead of 936
// (bench.py --track-order shuffled).  A counting sort of the previous positions by cell (64-px bands, 32-px columns inside a band: raster order of cells)
// gives the launch the locality of a raster-ordered track list whatever order the caller keeps; outputs stay at the caller's indices.  The order inside
// a cell is whatever the LDS atomics make it: it affects scheduling only, never a result.
#define KO_BAND_SHIFT 6
#define KO_COL_SHIFT 5
#define KO_BANDS 40
#define KO_COLS 128
#define KO_THREADS 256
__device__ __forceinline__ int klt_order_key(float x, float y)
{
    const int b = min(max((int)y >> KO_BAND_SHIFT, 0), KO_BANDS - 1), c = min(max((int)x >> KO_COL_SHIFT, 0), KO_COLS - 1);
    return b * KO_COLS + c;
}
__device__ void klt_order_tracks(const float* p0, int n, int* order)
{
    __shared__ int bins[KO_BANDS * KO_COLS];
    __shared__ int wsum[KO_THREADS / 64];
    const int tid = threadIdx.x;
    constexpr int PER = KO_BANDS * KO_COLS / KO_THREADS;
    for (int k = tid; k < KO_BANDS * KO_COLS; k += KO_THREADS) bins[k] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += KO_THREADS) {
        const float x = p0[2 * i], y = p0[2 * i + 1];
        atomicAdd(&bins[klt_order_key(x == x ? x : 0.f, y == y ? y : 0.f)], 1);
    }
    __syncthreads();
    // exclusive scan: a thread owns PER consecutive bins
    int loc[PER], tot = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { loc[k] = tot; tot += bins[tid * PER + k]; }
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
#pragma unroll
    for (int k = 0; k < PER; k++) bins[tid * PER + k] = base + loc[k];
    __syncthreads();
    for (int i = tid; i < n; i += KO_THREADS) {
        const float x = p0[2 * i], y = p0[2 * i + 1];
        order[atomicAdd(&bins[klt_order_key(x == x ? x : 0.f, y == y ? y : 0.f)], 1)] = i;
    }
}

__global__ __launch_bounds__(KO_THREADS) void k_klt_setup(StreamWS* ws_all, const SessStream* ss_all, const uint8_t* const* frames, vh_lk_params coarse,
                                                            vh_lk_params fine, int use_order)
{
    StreamWS& ws = ws_all[blockIdx.x];
    if (threadIdx.x == 0) klt_setup_descriptors(ws, ss_all, frames, coarse, fine, use_order);
    if (!use_order) return;
    __threadfence_block();
    __syncthreads();
    klt_order_tracks(ws.io.p0, ws.n, ws.bufs.order);
}

__device__ void klt_setup_descriptors(StreamWS& ws, const SessStream* ss_all, const uint8_t* const* frames, const vh_lk_params& coarse,
                                      const vh_lk_params& fine, int use_order)
{
    if (ss_all) {  // session mode: this frame's KLTmain call (vidExample.py:134) straight from the stream state
        const SessStream& S = ss_all[blockIdx.x];
        KltIO& o = ws.io;
        o.im = frames[blockIdx.x];
        o.im0 = S.im0;
        o.im0_small = S.small[1 - S.pp];
        o.im_small = S.small[S.pp];
        o.p0 = S.p_cur;
        o.n_ptr = &S.n_cur;
        o.n = 0;
        o.p_all = S.p_all;
        o.v = S.v;
        o.flags = const_cast<int*>(&S.klt_flags);
        o.w = S.w; o.h = S.h; o.stride = S.stride; o.stride0 = S.stride;
        o.reuse_prev_small = S.frame_i >= 1 ? 1 : 0;  // the previous step built the pyramid of what is now im0_small
        o.have_small = S.small_ready == S.frame_i + 1 ? 1 : 0;  // vh_session_ingest_bgr wrote this frame's quarter-scale image already
        o.coarse = coarse; o.fine = fine;
        o.fbt_coarse = 1.0f; o.fbt_fine = 0.3f;
        ws.pp = S.pp;
    }
    const KltIO& io = ws.io;
    const StreamBufs& B = ws.bufs;
    const int n = io.n_ptr ? *io.n_ptr : io.n;
    ws.n = n;
    ws.flags = 0;
    ws.order = use_order ? B.order : nullptr;
    const int dw = __double2int_rn(io.w * 0.25), dh = __double2int_rn(io.h * 0.25);
    const int cur = ws.pp & 1, prev = 1 - cur;
    uint8_t* small_cur = io.im_small ? io.im_small : B.small0[cur];
    const uint8_t* small_prev = io.im0_small ? io.im0_small : B.small0[prev];
    // resize table
    ws.rs_src[0] = ImgDesc{io.im, io.w, io.h, io.stride, 0};
    const bool have_small = ss_all != nullptr && io.have_small != 0;
    ws.rs_dst[0] = ImgDesc{small_cur, have_small ? 0 : dw, have_small ? 0 : dh, dw, 0};
    const bool need_prev = io.im0_small == nullptr;
    ws.rs_src[1] = ImgDesc{io.im0, io.w, io.h, io.stride0, 0};
    ws.rs_dst[1] = ImgDesc{B.small0[prev], need_prev ? dw : 0, need_prev ? dh : 0, dw, 0};
    // job A: LK on the quarter-scale pair, points scaled by 1/4, no backward pass
    LKJob& J = ws.lk;
    fill_pyramid(J.I, small_prev, dw, dh, dw, B.small_lv[prev], io.coarse.win, io.coarse.max_level);
    fill_pyramid(J.J, small_cur, dw, dh, dw, B.small_lv[cur], io.coarse.win, io.coarse.max_level);
    ws.pb[0] = PyrBuild{&J.I, io.reuse_prev_small ? 0 : 1, 0};
    ws.pb[1] = PyrBuild{&J.J, 1, 0};
    fill_lk_common(J, io.coarse, io.p0, nullptr, n);
    J.order = ws.order;
    J.p_out = B.p_small; J.v_out = B.v_small;
    J.fbt = -1.f;
    J.in_scale = 0.25f; J.in_off[0] = 0.f; J.in_off[1] = 0.f;
    J.out_mode = VH_OUT_SCALE; J.out_scale = 0.25f;
    J.stats = &ws.lk_stats[0][0][0];
    // RANSAC 1: inliers gate the status (KLT.py:116-117)
    RansacJob& R = ws.ransac;
    R.from = io.p0; R.to = B.p_small; R.valid = B.v_small; R.n_ptr = nullptr; R.n = n;
    R.min_valid = 0; R.gate_valid = 1;
    R.idx = B.idx; R.pairs = B.pairs; R.counts = B.counts; R.m_out = &ws.m; R.bound = &ws.rbound; R.M = ws.M; R.inl = B.inl; R.status = &ws.rstatus;
}

// ---- stage 1 -> 2 and stage 2 -> 3 glue: bodies in vh_glue_dev.hpp (shared with the fused RANSAC kernel, which runs them as its epilogue) -------------
__global__ __launch_bounds__(256) void k_klt_glue1(StreamWS* ws_all) { klt_glue1_body<256>(ws_all[blockIdx.x]); }
__global__ void k_klt_glue2(StreamWS* ws_all) { klt_glue2_body(ws_all[blockIdx.x]); }

static int launch_lk_profiled(vh_ctx* c, int stage, const void* tab, size_t st, int count, int win, hipStream_t s, int mn)
{
    const int rec = vh_prof_start(c, s, 1);
    int route = 0, tpw = 1;
    const int r = vh_launch_lk(tab, st, count, mn, win, s, &route, &tpw);
    if (c && stage >= 0 && stage < 3) { c->lk_route[stage] = route; c->lk_win[stage] = win; c->lk_tpw[stage] = tpw; }
    vh_prof_stop(c, rec, stage, s);
    return r;
}
#define VH_PROFILED(c, stage, s, ...)                 \
    do {                                              \
        const int rec_ = vh_prof_start((c), (s));     \
        __VA_ARGS__;                                  \
        vh_prof_stop((c), rec_, (stage), (s));        \
    } while (0)

static std::atomic<int> g_klt_order{getenv("VH_KLT_ORDER") ? atoi(getenv("VH_KLT_ORDER")) : -1};  // PROCESS-WIDE test hook: 1 always, 0 never, -1 by load
extern "C" VH_API void vh_debug_klt_order(int mode) { g_klt_order.store(mode, std::memory_order_relaxed); }

int vh_run_klt_main(vh_ctx* c, int slot, int count, hipStream_t s, const vh_lk_params& coarse, const vh_lk_params& fine, const SessStream* sess,
                    const uint8_t* const* frames, int n_max)
{
    const int mn = n_max > 0 && n_max < c->max_pts ? n_max : c->max_pts;  // launch extent over the tracks (the kernels read the real count)
    StreamWS* ws = c->d_ws + slot;
    const size_t st = sizeof(StreamWS);
    const int lvl_c = min(coarse.max_level, VH_MAX_LEVELS - 1), lvl_f = min(fine.max_level, VH_MAX_LEVELS - 1);
    // spatial launch order of the tracks: pays from the load at which a stream's pyramids no longer sit in every L2 anyway (8 streams: no difference measured)
    const int order_mode = g_klt_order.load(std::memory_order_relaxed);
    const int use_order = order_mode >= 0 ? order_mode : ((long long)count * mn >= 24000 ? 1 : 0);
    hipLaunchKernelGGL(k_klt_setup, dim3(count), dim3(use_order ? KO_THREADS : 64), 0, s, ws, sess ? sess + slot : nullptr, frames, coarse, fine, use_order);
    VH_PROFILED(c, VH_PROF_RESIZE, s, vh_launch_resize_quarter(&ws->rs_src[0], &ws->rs_dst[0], st, 2, count, c->sw, c->sh, s));
    VH_PROFILED(c, VH_PROF_PYR, s, for (int l = 0; l < lvl_c; l++) vh_launch_pyr_down_ws(&ws->pb[0], st, count, l, c->sw, c->sh, s));
    int r = launch_lk_profiled(c, 0, &ws->lk, st, count, coarse.win, s, mn);
    if (r) return vh_fail(r, "vh_launch_lk failed (window too large for LDS?)");
    // RANSAC + the glue that follows it: ONE launch when the fused one-workgroup-per-stream kernel serves the problem (it runs the glue as its epilogue),
    // else the three RANSAC launches and the glue kernel (which does the zero-padded shifted crop itself when one is needed)
    bool glued = false;
    VH_PROFILED(c, VH_PROF_RANSAC, s, glued = vh_launch_ransac(&ws->ransac, st, count, mn, s, ws, 1));
    if (!glued) hipLaunchKernelGGL(k_klt_glue1, dim3(count), dim3(256), 0, s, ws);
    VH_PROFILED(c, VH_PROF_PYR, s, for (int l = 0; l < lvl_c; l++) vh_launch_pyr_down_ws(&ws->pb[0], st, count, l, c->max_w, c->max_h, s));
    r = launch_lk_profiled(c, 1, &ws->lk, st, count, coarse.win, s, mn);
    if (r) return vh_fail(r, "vh_launch_lk failed");
    VH_PROFILED(c, VH_PROF_RANSAC, s, glued = vh_launch_ransac(&ws->ransac, st, count, mn, s, ws, 2));
    if (!glued) hipLaunchKernelGGL(k_klt_glue2, dim3(count), dim3(64), 0, s, ws);
    VH_PROFILED(c, VH_PROF_WARP, s, vh_launch_roi_warp(&ws->warp, st, count, c->max_w, c->max_h, s));
    if (lvl_f > 0) VH_PROFILED(c, VH_PROF_PYR, s, for (int l = 0; l < lvl_f; l++) vh_launch_pyr_down_ws(&ws->pb[0], st, count, l, c->max_w, c->max_h, s));
    r = launch_lk_profiled(c, 2, &ws->lk, st, count, fine.win, s, mn);
    if (r) return vh_fail(r, "vh_launch_lk failed");
    VH_LAUNCH_CHECK();
    return 0;
}

// ---- optional HIP-event timing of the LK launches + iteration statistics (bench.py's roofline leg) ---------------
extern "C" VH_API int vh_profile_begin(vh_ctx* c, int max_launches)
{
    if (!c || max_launches < 1) return vh_fail(-1, "vh_profile_begin: bad arguments");
    if (c->prof_cap < max_launches) {
        for (int k = 0; k < 2 * c->prof_cap; k++) (void)hipEventDestroy(c->prof_ev[k]);
        delete[] c->prof_ev;
        delete[] c->prof_stage;
        c->prof_ev = new hipEvent_t[2 * max_launches];
        c->prof_stage = new int[max_launches];
        for (int k = 0; k < 2 * max_launches; k++) VH_CHECK(hipEventCreate(&c->prof_ev[k]));
        c->prof_cap = max_launches;
    }
    c->prof_n = 0;
    c->prof_dropped = 0;
    c->prof_on = c->prof_light ? 1 : 2;
    VH_CHECK(hipDeviceSynchronize());
    for (int b = 0; b < c->batch; b++) VH_CHECK(hipMemset(c->d_ws[b].lk_stats, 0, sizeof(c->d_ws[b].lk_stats)));
    return 0;
}

// ms_sum[3], launches[3], iters[3], setups[3]: per KLTmain stage (0: quarter scale, 1: coarse ROI, 2: fine), summed over streams
extern "C" VH_API int vh_profile_end(vh_ctx* c, double* ms_sum, int* launches, unsigned long long* iters, unsigned long long* setups)
{
    if (!c) return vh_fail(-1, "null ctx");
    c->prof_on = 0;
    VH_CHECK(hipDeviceSynchronize());
    if (c->prof_dropped) return vh_fail(-5, "vh_profile_end: the record table given to vh_profile_begin was too small, launches went unrecorded (size it for ~22 records per frame step + 5 per BA iteration, or call vh_profile_detail(ctx, 0))");
    for (int k = 0; k < 3; k++) { ms_sum[k] = 0; launches[k] = 0; iters[k] = 0; setups[k] = 0; }
    for (int k = 0; k < c->prof_n; k++) {
        float ms = 0.f;
        VH_CHECK(hipEventElapsedTime(&ms, c->prof_ev[2 * k], c->prof_ev[2 * k + 1]));
        if (c->prof_stage[k] > 2) continue;  // the other stages: vh_profile_end_stages
        ms_sum[c->prof_stage[k]] += ms;
        launches[c->prof_stage[k]]++;
    }
    for (int b = 0; b < c->batch; b++) {
        static_assert(sizeof(c->d_ws[b].lk_stats) == sizeof(unsigned long long) * 3 * VH_LK_STAT_SLOTS * 16, "lk_stats layout");
        std::vector<unsigned long long> st((size_t)3 * VH_LK_STAT_SLOTS * 16);
        VH_CHECK(hipMemcpy(st.data(), c->d_ws[b].lk_stats, sizeof(c->d_ws[b].lk_stats), hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; k++)
            for (int q = 0; q < VH_LK_STAT_SLOTS; q++) { iters[k] += st[((size_t)k * VH_LK_STAT_SLOTS + q) * 16]; setups[k] += st[((size_t)k * VH_LK_STAT_SLOTS + q) * 16 + 1]; }
    }
    return 0;
}

// detail of the NEXT vh_profile_begin: 1 (default) times every stage, 0 only the three LK launches of a step (6 event records per step instead of
// ~22: the choice for single-stream latency runs, where an event record between two 5 us kernels is not free)
extern "C" VH_API int vh_profile_detail(vh_ctx* c, int all_stages)
{
    if (!c) return vh_fail(-1, "null ctx");
    c->prof_light = all_stages ? 0 : 1;
    return 0;
}

// every profiled stage (VH_PROF_* in vh_ws.hpp): ms_sum[nstages], launches[nstages]; call after (or instead of) vh_profile_end
extern "C" VH_API int vh_profile_end_stages(vh_ctx* c, int nstages, double* ms_sum, int* launches)
{
    if (!c || nstages < 1 || !ms_sum || !launches) return vh_fail(-1, "vh_profile_end_stages: bad arguments");
    c->prof_on = 0;
    VH_CHECK(hipDeviceSynchronize());
    if (c->prof_dropped) return vh_fail(-5, "vh_profile_end_stages: the record table given to vh_profile_begin was too small, launches went unrecorded");
    for (int k = 0; k < nstages; k++) { ms_sum[k] = 0; launches[k] = 0; }
    for (int k = 0; k < c->prof_n; k++) {
        float ms = 0.f;
        VH_CHECK(hipEventElapsedTime(&ms, c->prof_ev[2 * k], c->prof_ev[2 * k + 1]));
        const int st = c->prof_stage[k];
        if (st >= 0 && st < nstages) { ms_sum[st] += ms; launches[st]++; }
    }
    return 0;
}

extern "C" VH_API int vh_ba_graph_replay(vh_ctx* c, int on)
{
    if (!c) return vh_fail(-1, "null ctx");
    c->ba_graph_on = on ? 1 : 0;
    return 0;
}

// ROI of the last KLTmain call of every stream (x0, x1, y0, y1), host copy: the algorithmic bytes of the ROI-sized stages (bench.py's per-kernel rows)
extern "C" VH_API int vh_klt_rois(vh_ctx* c, int* roi_host)
{
    if (!c || !roi_host) return vh_fail(-1, "vh_klt_rois: bad arguments");
    VH_CHECK(hipDeviceSynchronize());
    for (int b = 0; b < c->batch; b++) VH_CHECK(hipMemcpy(roi_host + 4 * b, c->d_ws[b].roi, 4 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// which kernel each of the three LK launches of the last KLTmain / session step took (the launcher's own decision, vh_lk_route): ids as
// vh_debug_force_generic_lk, names as in DESIGN.md / rocprofv3 traces; names_host: 3 x 32 chars (may be null)
// launch slots per workgroup of the same three launches (1 unless the one-wavefront LDS-staged kernel looped over consecutive slots, launch_lk3); a
// stateless vh_pyr_lk call reports through entry 0, like its iteration counters
extern "C" VH_API int vh_profile_lk_tpw(vh_ctx* c, int* tpw_host)
{
    if (!c || !tpw_host) return vh_fail(-1, "vh_profile_lk_tpw: bad arguments");
    for (int k = 0; k < 3; k++) tpw_host[k] = c->lk_tpw[k];
    return 0;
}

void vh_lk3_set_tpw(int n);
extern "C" VH_API void vh_debug_lk3_tpw(int n) { vh_lk3_set_tpw(n); }

extern "C" VH_API int vh_profile_lk_routes(vh_ctx* c, int* routes_host, char* names_host)
{
    if (!c || !routes_host) return vh_fail(-1, "vh_profile_lk_routes: bad arguments");
    for (int k = 0; k < 3; k++) {
        routes_host[k] = c->lk_route[k];
        if (names_host) snprintf(names_host + 32 * k, 32, "%s", vh_lk_route_name(c->lk_route[k], c->lk_win[k]));
    }
    return 0;
}

extern "C" VH_API int vh_klt_main(vh_ctx* c, int slot, const uint8_t* im, const uint8_t* im0, const uint8_t* im0_small, int w, int h,
                                  int stride, int stride0, const float* p0, int n, const vh_lk_params* coarse, const vh_lk_params* fine,
                                  float* p_all, uint8_t* v, uint8_t* im_small, int* flags, void* stream)
{
    if (!c || slot < 0 || slot >= c->batch) return vh_fail(-1, "vh_klt_main: bad slot");
    if (w > c->max_w || h > c->max_h || n > c->max_pts || n < 0) return vh_fail(-1, "vh_klt_main: frame or point count exceeds the workspace");
    if (!coarse || !fine || coarse->win < 3 || fine->win < 3 || coarse->max_level < 0 || fine->max_level < 0)
        return vh_fail(-1, "vh_klt_main: bad LK parameters (need win >= 3, max_level >= 0)");
    if (w < 4 || h < 4 || stride < w || stride0 < w) return vh_fail(-1, "vh_klt_main: frames must be at least 4 x 4 with row strides >= width");
    KltIO io;
    memset(&io, 0, sizeof(io));
    io.im = im; io.im0 = im0; io.im0_small = im0_small; io.p0 = p0; io.n_ptr = nullptr; io.p_all = p_all; io.v = v;
    io.im_small = im_small; io.flags = flags; io.w = w; io.h = h; io.stride = stride; io.stride0 = stride0; io.n = n;
    io.reuse_prev_small = 0; io.coarse = *coarse; io.fine = *fine; io.fbt_coarse = 1.0f; io.fbt_fine = 0.3f;
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    VH_CHECK(vh_store(&c->d_ws[slot].io, io, s));
    return vh_run_klt_main(c, slot, 1, s, *coarse, *fine, nullptr, nullptr, n);
}

extern "C" VH_API int vh_klt_stage_ptrs(vh_ctx* c, int slot, vh_klt_stages* out)
{
    if (!c || !out || slot < 0 || slot >= c->batch) return vh_fail(-1, "vh_klt_stage_ptrs: bad arguments");
    const StreamBufs& B = c->h_bufs[slot];
    StreamWS* ws = c->d_ws + slot;
    out->p_small = B.p_small; out->v_small = B.v_small; out->t_trans = ws->t_trans; out->roi = ws->roi;
    out->p_coarse = B.p_coarse; out->v_coarse = B.v_coarse; out->t23 = ws->M; out->warped = B.warp; out->flags = &ws->flags;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// stateless image entry points (use slot 0 of the workspace for their descriptors)
// ---------------------------------------------------------------------------------------------------------------
extern "C" VH_API int vh_resize_quarter(vh_ctx* c, const uint8_t* src, int w, int h, int stride, uint8_t* dst, void* stream)
{
    if (!c) return vh_fail(-1, "null ctx");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    const int dw = (int)lrint(w * 0.25), dh = (int)lrint(h * 0.25);
    StreamWS* ws = c->d_ws;
    VH_CHECK(vh_store(&ws->rs_src[0], ImgDesc{src, w, h, stride, 0}, s));
    VH_CHECK(vh_store(&ws->rs_dst[0], ImgDesc{dst, dw, dh, dw, 0}, s));
    vh_launch_resize_quarter(&ws->rs_src[0], &ws->rs_dst[0], sizeof(StreamWS), 1, 1, dw, dh, s);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_resize_nearest(vh_ctx* c, const uint8_t* src, int w, int h, int stride, double fx, double fy, uint8_t* dst, int dst_stride,
                                        void* stream)
{
    if (!c || w < 1 || h < 1 || !(fx > 0) || !(fy > 0)) return vh_fail(-1, "vh_resize_nearest: bad arguments");
    const int dw = (int)lrint(w * fx), dh = (int)lrint(h * fy);
    if (dw < 1 || dh < 1 || dst_stride < dw) return vh_fail(-1, "vh_resize_nearest: bad output size");
    vh_launch_resize_nearest(src, w, h, (size_t)stride, dst, dw, dh, (size_t)dst_stride, 1.0 / fx, 1.0 / fy, (hipStream_t)stream);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_bgr2gray(vh_ctx* c, const uint8_t* bgr, int w, int h, int stride_bytes, uint8_t* gray, int gray_stride, void* stream)
{
    if (!c || w < 1 || h < 1) return vh_fail(-1, "vh_bgr2gray: bad arguments");
    vh_launch_bgr2gray(bgr, w, h, (size_t)stride_bytes, gray, (size_t)gray_stride, (hipStream_t)stream);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_ingest_bgr(vh_ctx* c, const uint8_t* bgr, int w, int h, int stride_bytes, uint8_t* gray, int gray_stride, uint8_t* small, void* stream)
{
    if (!c || !bgr || !gray || w < 1 || h < 1 || stride_bytes < 3 * w || gray_stride < w) return vh_fail(-1, "vh_ingest_bgr: bad arguments");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    IngestJob J;
    memset(&J, 0, sizeof(J));
    J.bgr = bgr; J.gray = gray; J.small = small; J.w = w; J.h = h; J.bgr_stride = stride_bytes; J.gray_stride = gray_stride;
    J.dw = (int)lrint(w * 0.25); J.dh = (int)lrint(h * 0.25); J.small_stride = J.dw;
    static_assert(sizeof(IngestJob) <= sizeof(LKJob), "IngestJob must fit in the LKJob slot");
    IngestJob* d = reinterpret_cast<IngestJob*>(&c->d_ws[0].lk);  // parked like every stateless call's descriptor
    VH_CHECK(vh_store(d, J, s));
    vh_launch_ingest_bgr(d, 1, w, h, s);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_pyr_down(vh_ctx* c, const uint8_t* src, int w, int h, int stride, uint8_t* dst, void* stream)
{
    if (!c) return vh_fail(-1, "null ctx");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    StreamWS* ws = c->d_ws;
    PyrDesc P;
    memset(&P, 0, sizeof(P));
    P.nlevels = 2;
    P.lv[0] = ImgDesc{src, w, h, stride, 0};
    P.lv[1] = ImgDesc{dst, (w + 1) / 2, (h + 1) / 2, (w + 1) / 2, 0};
    VH_CHECK(vh_store(&ws->lk.I, P, s));
    VH_CHECK(vh_store(&ws->pb[0], PyrBuild{&ws->lk.I, 1, 0}, s));
    VH_CHECK(vh_store(&ws->pb[1], PyrBuild{nullptr, 0, 0}, s));
    vh_launch_pyr_down_ws(&ws->pb[0], sizeof(StreamWS), 1, 0, w, h, s);
    VH_LAUNCH_CHECK();
    return 0;
}

static int run_warp(vh_ctx* c, const WarpJob& job, hipStream_t s)
{
    StreamWS* ws = c->d_ws;
    VH_CHECK(vh_store(&ws->warp, job, s));
    vh_launch_roi_warp(&ws->warp, sizeof(StreamWS), 1, job.x1 - job.x0, job.y1 - job.y0, s);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_remap_affine(vh_ctx* c, const uint8_t* im, int w, int h, int stride, const float* T, int x0, int x1, int y0,
                                      int y1, uint8_t* dst, void* stream)
{
    if (!c || x1 <= x0 || y1 <= y0) return vh_fail(-1, "vh_remap_affine: bad arguments");
    WarpJob J;
    memset(&J, 0, sizeof(J));
    J.src = ImgDesc{im, w, h, stride, 0}; J.dst = dst; J.dst_stride = x1 - x0; J.mode = 1;
    J.x0 = x0; J.x1 = x1; J.y0 = y0; J.y1 = y1;
    for (int k = 0; k < 6; k++) J.T[k] = T[k];
    return run_warp(c, J, vh_ctx_bind(c, stream));
}

extern "C" VH_API int vh_crop_shift(vh_ctx* c, const uint8_t* im, int w, int h, int stride, int x0, int x1, int y0, int y1, int dx, int dy,
                                    uint8_t* dst, void* stream)
{
    if (!c || x1 <= x0 || y1 <= y0) return vh_fail(-1, "vh_crop_shift: bad arguments");
    WarpJob J;
    memset(&J, 0, sizeof(J));
    J.src = ImgDesc{im, w, h, stride, 0}; J.dst = dst; J.dst_stride = x1 - x0; J.mode = 0;
    J.x0 = x0; J.x1 = x1; J.y0 = y0; J.y1 = y1; J.dx = dx; J.dy = dy;
    return run_warp(c, J, vh_ctx_bind(c, stream));
}

__global__ __launch_bounds__(256) void k_bounding_rect(const float* p, int n, int imw, int imh, int bx, int by, int* roi)
{
    __shared__ float sh_f[4];
    float mnx = 3.0e38f, mny = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = p[2 * i], y = p[2 * i + 1];
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
    }
    mnx = block_min_f<4>(mnx, sh_f, false); mny = block_min_f<4>(mny, sh_f, false);
    mxx = block_min_f<4>(mxx, sh_f, true);  mxy = block_min_f<4>(mxy, sh_f, true);
    if (threadIdx.x == 0) {
        int x0 = vh_floor(mnx), y0 = vh_floor(mny);
        const int bw = vh_floor(mxx) - x0 + 1, bh = vh_floor(mxy) - y0 + 1;
        int x1 = x0 + bw + bx, y1 = y0 + bh + by;
        x0 -= bx; y0 -= by;
        roi[0] = max(x0, 1); roi[1] = min(x1, imw); roi[2] = max(y0, 1); roi[3] = min(y1, imh);
    }
}

extern "C" VH_API int vh_bounding_rect(vh_ctx* c, const float* p, int n, int imw, int imh, int bx, int by, int* roi_out, void* stream)
{
    if (!c || n < 1) return vh_fail(-1, "vh_bounding_rect: bad arguments");
    hipLaunchKernelGGL(k_bounding_rect, dim3(1), dim3(256), 0, (hipStream_t)stream, p, n, imw, imh, bx, by, roi_out);
    VH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// stateless tracker entry points
// ---------------------------------------------------------------------------------------------------------------
static void host_fill_pyramid(PyrDesc& P, const uint8_t* lv0, int w, int h, int stride, uint8_t* const* lvbuf, int win, int max_level)
{
    int n = 0;
    for (int level = 0; level <= max_level && level < VH_MAX_LEVELS; level++) {
        ImgDesc& d = P.lv[level];
        if (level == 0) { d.p = lv0; d.w = w; d.h = h; d.stride = stride; d.pad = 0; }
        else if (VH_LV_PADDED(w, h)) { d.stride = VH_LV_STRIDE(w); d.p = lvbuf[level] + (size_t)VH_LV_PAD * d.stride + VH_LV_PAD; d.w = w; d.h = h; d.pad = VH_LV_PAD; }
        else { d.p = lvbuf[level]; d.w = w; d.h = h; d.stride = (w + 3) & ~3; d.pad = 0; }  // dword rows: k_pyr_down stores packed dwords (byte stores on a 766-pixel pitch cost it 2x)
        n = level + 1;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win || h <= win) break;
    }
    P.nlevels = n;
}

extern "C" VH_API int vh_pyr_lk(vh_ctx* c, const uint8_t* im1, const uint8_t* im2, int w, int h, int stride1, int stride2, const float* p1,
                                int n, const vh_lk_params* lk, float fbt, float* p2, uint8_t* v, float* err, float* fbe, void* stream)
{
    if (!c || !lk || lk->win < 3 || lk->max_level < 0 || w < 4 || h < 4 || stride1 < w || stride2 < w || n < 0)
        return vh_fail(-1, "vh_pyr_lk: bad arguments (need win >= 3, max_level >= 0, w, h >= 4, strides >= w)");
    if (w > c->max_w || h > c->max_h || n > c->max_pts) return vh_fail(-1, "vh_pyr_lk: image or point count exceeds the workspace");
    if (n <= 0) return 0;
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    const StreamBufs& B = c->h_bufs[0];
    LKJob J;
    memset(&J, 0, sizeof(J));
    host_fill_pyramid(J.I, im1, w, h, stride1, B.roi_lv[0], lk->win, lk->max_level);
    host_fill_pyramid(J.J, im2, w, h, stride2, B.roi_lv[1], lk->win, lk->max_level);
    J.p_in = p1; J.p_out = p2; J.v_out = v; J.err_out = err; J.fbe_out = fbe; J.n = n;
    J.win = lk->win; J.max_level = lk->max_level;
    J.max_count = lk->max_count < 0 ? 0 : (lk->max_count > 100 ? 100 : lk->max_count);
    double e = lk->eps < 0 ? 0 : (lk->eps > 10 ? 10 : lk->eps);
    J.eps2 = e * e;
    J.fbt = fbt;
    J.in_scale = 1.f; J.out_mode = VH_OUT_SCALE; J.out_scale = 1.f;
    StreamWS* ws = c->d_ws;
    // Newton iterations / template set-ups of this call, ONLY between vh_profile_begin (zeroes them) and vh_profile_end (reads them, as stage 0): an
    // unprofiled call issues no statistics atomics at all
    J.stats = c->prof_on ? &ws->lk_stats[0][0][0] : nullptr;
    VH_CHECK(vh_store(&ws->lk, J, s));
    VH_CHECK(vh_store(&ws->pb[0], PyrBuild{&ws->lk.I, 1, 0}, s));
    VH_CHECK(vh_store(&ws->pb[1], PyrBuild{&ws->lk.J, 1, 0}, s));
    const int lv = lk->max_level < VH_MAX_LEVELS - 1 ? lk->max_level : VH_MAX_LEVELS - 1;
    for (int l = 0; l < lv; l++) vh_launch_pyr_down_ws(&ws->pb[0], sizeof(StreamWS), 1, l, w, h, s);
    int r = vh_launch_lk(&ws->lk, sizeof(StreamWS), 1, n, lk->win, s, &c->lk_route[0], &c->lk_tpw[0]);
    c->lk_win[0] = lk->win;
    if (r) return vh_fail(r, "vh_launch_lk failed (window too large for LDS?)");
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_ransac_affine(vh_ctx* c, const float* from, const float* to, const uint8_t* valid, int n, double* M, uint8_t* inl,
                                       int* status, void* stream)
{
    if (!c || n < 0 || n > c->max_pts) return vh_fail(-1, "vh_ransac_affine: bad arguments");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    const StreamBufs& B = c->h_bufs[0];
    RansacJob R;
    memset(&R, 0, sizeof(R));
    R.from = from; R.to = to; R.valid = const_cast<uint8_t*>(valid ? valid : B.v_all); R.n = n; R.min_valid = 0; R.gate_valid = 0;
    R.idx = B.idx; R.pairs = B.pairs; R.counts = B.counts; R.m_out = &c->d_ws[0].m; R.bound = &c->d_ws[0].rbound; R.M = M; R.inl = inl; R.status = status;
    VH_CHECK(vh_store(&c->d_ws[0].ransac, R, s));
    vh_launch_ransac(&c->d_ws[0].ransac, sizeof(StreamWS), 1, n, s);
    VH_LAUNCH_CHECK();
    return 0;
}

// KLTregional (KLT.py:55-95) as one device-side pipeline: bbox -> ROI -> crop/remap -> LK fwd/bwd -> map back
struct RegionalIO {
    const uint8_t* im0;
    const uint8_t* im;
    const float* p0;
    float* p_out;
    uint8_t* v_out;
    int* roi_out;
    int w, h, stride0, stride, n, translate;
    vh_lk_params lk;
    float fbt;
    float T[6];
};

__global__ __launch_bounds__(256) void k_regional_setup(StreamWS* ws_p, RegionalIO io)
{
    StreamWS& ws = *ws_p;
    const StreamBufs& B = ws.bufs;
    __shared__ float sh_f[4];
    float mnx = 3.0e38f, mny = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f;
    for (int i = threadIdx.x; i < io.n; i += 256) {
        const float x = io.p0[2 * i], y = io.p0[2 * i + 1];
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
    }
    mnx = block_min_f<4>(mnx, sh_f, false); mny = block_min_f<4>(mny, sh_f, false);
    mxx = block_min_f<4>(mxx, sh_f, true);  mxy = block_min_f<4>(mxy, sh_f, true);
    if (threadIdx.x != 0) return;
    int x0 = vh_floor(mnx), y0 = vh_floor(mny);
    const int bw = vh_floor(mxx) - x0 + 1, bh = vh_floor(mxy) - y0 + 1;
    int x1 = min(x0 + bw + 50, io.w), y1 = min(y0 + bh + 50, io.h);
    x0 = max(x0 - 50, 1); y0 = max(y0 - 50, 1);
    const int rw = max(x1 - x0, 0), rh = max(y1 - y0, 0);
    if (io.roi_out) { io.roi_out[0] = x0; io.roi_out[1] = x1; io.roi_out[2] = y0; io.roi_out[3] = y1; }
    ws.roi[0] = x0; ws.roi[1] = x1; ws.roi[2] = y0; ws.roi[3] = y1;
    WarpJob& W = ws.warp;
    W.src = ImgDesc{io.im, io.w, io.h, io.stride, 0};
    W.dst = B.warp; W.dst_stride = (rw + 3) & ~3;  // dword rows: the warp kernel stores packed dwords
    W.x0 = x0; W.x1 = x1; W.y0 = y0; W.y1 = y1;
    LKJob& J = ws.lk;
    fill_pyramid(J.I, io.im0 + (size_t)y0 * io.stride0 + x0, rw, rh, io.stride0, B.roi_lv[0], io.lk.win, io.lk.max_level);
    fill_lk_common(J, io.lk, io.p0, nullptr, io.n);
    J.p_out = io.p_out; J.v_out = io.v_out; J.fbt = io.fbt;
    J.in_scale = 1.f; J.in_off[0] = (float)x0; J.in_off[1] = (float)y0;
    if (io.translate) {
        const int dx = (int)io.T[4], dy = (int)io.T[5];
        W.dx = dx; W.dy = dy;
        const bool inside = x0 + dx >= 0 && x1 + dx <= io.w && y0 + dy >= 0 && y1 + dy <= io.h;
        W.mode = inside ? -1 : 0;
        if (inside) fill_pyramid(J.J, io.im + (ptrdiff_t)(y0 + dy) * io.stride + (x0 + dx), rw, rh, io.stride, B.roi_lv[1], io.lk.win, io.lk.max_level);
        else fill_pyramid(J.J, B.warp, rw, rh, (rw + 3) & ~3, B.roi_lv[1], io.lk.win, io.lk.max_level);
        J.out_mode = VH_OUT_TRANSLATE; J.out_off[0] = (float)dx; J.out_off[1] = (float)dy;
    } else {
        W.mode = 1;
        for (int k = 0; k < 6; k++) { W.T[k] = io.T[k]; J.T[k] = io.T[k]; }
        fill_pyramid(J.J, B.warp, rw, rh, (rw + 3) & ~3, B.roi_lv[1], io.lk.win, io.lk.max_level);
        J.out_mode = VH_OUT_AFFINE;
    }
    ws.pb[0] = PyrBuild{&J.I, 1, 0};
    ws.pb[1] = PyrBuild{&J.J, 1, 0};
}

extern "C" VH_API int vh_klt_regional(vh_ctx* c, const uint8_t* im0, const uint8_t* im, int w, int h, int stride0, int stride, const float* p0,
                                      int n, const float* T_host, const vh_lk_params* lk, float fbt, int translate, float* p_out,
                                      uint8_t* v_out, int* roi_out, void* stream)
{
    if (!c || !lk || !T_host || lk->win < 3 || lk->max_level < 0 || n < 1 || w < 4 || h < 4 || stride0 < w || stride < w)
        return vh_fail(-1, "vh_klt_regional: bad arguments (need win >= 3, max_level >= 0, w, h >= 4, strides >= w)");
    if (w > c->max_w || h > c->max_h || n > c->max_pts) return vh_fail(-1, "vh_klt_regional: image or point count exceeds the workspace");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    RegionalIO io;
    memset(&io, 0, sizeof(io));
    io.im0 = im0; io.im = im; io.p0 = p0; io.p_out = p_out; io.v_out = v_out; io.roi_out = roi_out;
    io.w = w; io.h = h; io.stride0 = stride0; io.stride = stride; io.n = n; io.translate = translate; io.lk = *lk; io.fbt = fbt;
    for (int k = 0; k < 6; k++) io.T[k] = T_host[k];
    StreamWS* ws = c->d_ws;
    hipLaunchKernelGGL(k_regional_setup, dim3(1), dim3(256), 0, s, ws, io);
    vh_launch_roi_warp(&ws->warp, sizeof(StreamWS), 1, w, h, s);
    const int lv = lk->max_level < VH_MAX_LEVELS - 1 ? lk->max_level : VH_MAX_LEVELS - 1;
    for (int l = 0; l < lv; l++) vh_launch_pyr_down_ws(&ws->pb[0], sizeof(StreamWS), 1, l, w, h, s);
    int r = vh_launch_lk(&ws->lk, sizeof(StreamWS), 1, n, lk->win, s);
    if (r) return vh_fail(r, "vh_launch_lk failed (window too large for LDS?)");
    VH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// NLS / MSV entry points
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_store_doubles(double* dst, const double* src_dummy, int n, double v0, double v1, double v2, double v3, double v4, double v5,
                                double v6, double v7, double v8, double v9, double v10, double v11)
{
    const double v[12] = {v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11};
    if (threadIdx.x < n) dst[threadIdx.x] = v[threadIdx.x];
}
static int store_doubles(double* dst, const double* host, int n, hipStream_t s)
{
    double v[12] = {0};
    for (int k = 0; k < n && k < 12; k++) v[k] = host[k];
    hipLaunchKernelGGL(k_store_doubles, dim3(1), dim3(64), 0, s, dst, (const double*)nullptr, n, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8],
                       v[9], v[10], v[11]);
    return (int)hipGetLastError();
}

struct PoseSlot {  // lives in the workspace arena right behind d_small (one per stateless call)
    PoseJob job;
};

extern "C" VH_API int vh_pose(vh_ctx* c, const double* K, const float* p, const double* pw, int n, const double* x0, const double* R,
                              int findR, float* t_out, double* R_out, double* res_out, double* p_proj, int* info, void* stream)
{
    if (!c || !K || !x0 || !R || n < 0) return vh_fail(-1, "vh_pose: bad arguments");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    PoseJob J;
    memset(&J, 0, sizeof(J));
    for (int k = 0; k < 9; k++) { J.K[k] = (double)K[k]; J.R[k] = R[k]; }
    for (int k = 0; k < 6; k++) J.x0[k] = x0[k];
    J.p = p; J.pw = pw; J.n = n; J.mode = findR ? 1 : 0;
    J.t_out = t_out; J.R_out = R_out; J.res_out = res_out; J.p_proj = p_proj; J.info_out = info;
    // the job descriptor is parked in the LKJob area of slot 0 (never live at the same time on one stream)
    static_assert(sizeof(PoseJob) <= sizeof(LKJob), "PoseJob must fit in the LKJob slot");
    PoseJob* d = reinterpret_cast<PoseJob*>(&c->d_ws[0].lk);
    VH_CHECK(vh_store(d, J, s));
    vh_launch_pose(d, sizeof(PoseJob), 1, J.mode, n, s);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_world2image(vh_ctx* c, const double* C_host, const double* pw, int n, double* out, void* stream)
{
    if (!c || !C_host) return vh_fail(-1, "vh_world2image: bad arguments");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    int r = store_doubles(c->d_small, C_host, 12, s);
    if (r) return r;
    vh_launch_world2image(c->d_small, pw, n, out, s);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_image2world(vh_ctx* c, const double* Hi_host, const double* p, int n, double* out, void* stream)
{
    if (!c || !Hi_host) return vh_fail(-1, "vh_image2world: bad arguments");
    VH_BIND(c, stream);
    hipStream_t s = bound_.s;
    int r = store_doubles(c->d_small + 16, Hi_host, 9, s);
    if (r) return r;
    vh_launch_image2world(c->d_small + 16, p, n, out, s);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_pixel2uvec(vh_ctx* c, double cx, double cy, double f, const double* p, int n, double* out, void* stream)
{
    if (!c) return vh_fail(-1, "null ctx");
    vh_launch_pixel2uvec(cx, cy, f, p, n, out, (hipStream_t)stream);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_pixel2uvec_f32(vh_ctx* c, float cx, float cy, float f, const float* p, int n, float* out, void* stream)
{
    if (!c) return vh_fail(-1, "null ctx");
    vh_launch_pixel2uvec_f32(cx, cy, f, p, n, out, (hipStream_t)stream);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_two_view_intercept(vh_ctx* c, const double* A, const double* U, int nf, int nv, double* out, void* stream)
{
    if (!c || nf < 2) return vh_fail(-1, "vh_two_view_intercept: nf must be >= 2");
    vh_launch_two_view(A, U, nf, nv, out, (hipStream_t)stream);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_n_view_intercept(vh_ctx* c, const double* A, const double* U, int nf, int nv, double* out, void* stream)
{
    if (!c || nf < 2) return vh_fail(-1, "vh_n_view_intercept: nf must be >= 2");
    vh_launch_n_view(A, U, nf, nv, out, (hipStream_t)stream);
    VH_LAUNCH_CHECK();
    return 0;
}

extern "C" VH_API int vh_msv1_t(vh_ctx* c, const double* K, const float* P, const float* B, const int* ids, int ng, int N0, int nhist, int ii,
                                int f32_rays, double* U_scratch, float* x_out, double* b0, int* info, void* stream)
{
    if (!c || !K || ii < 1 || ii + 1 > nhist || ii + 1 > 2048) return vh_fail(-1, "vh_msv1_t: need 2 <= ii+1 <= min(nhist, 2048)");
    MsvJob J;
    memset(&J, 0, sizeof(J));
    for (int k = 0; k < 9; k++) J.K[k] = (double)K[k];
    J.P = P; J.P_rs = (size_t)N0 * nhist; J.P_ts = (size_t)nhist; J.P_fs = 1;  // the reference's [5, N0, nhist]
    J.B = B; J.ids = ids; J.ng = ng; J.N0 = N0; J.nhist = nhist; J.nf = ii + 1; J.max_iter = 1000; J.f32_rays = f32_rays;
    J.U = U_scratch; J.b0 = b0; J.x_out = x_out; J.info_out = info;
    vh_launch_msv1(J, (hipStream_t)stream);
    VH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// bundle adjustment entry points
// ---------------------------------------------------------------------------------------------------------------
// partials of the reduced system: one workgroup per ~16 points, at most 256 (512 / 1024 measured 20 % slower at C5: the reduction grows); with many
// cameras fewer, so that the partial systems of one window stay below 256 MB
static int ba_parts(int nt, int nc)
{
    static const int cap_env = [] { const char* e = getenv("VH_BA_PARTS"); return e ? atoi(e) : 0; }();  // experiment switch
    const int pmax = cap_env > 0 ? cap_env : 256;
    int p = nt / 16;
    p = p < 1 ? 1 : (p > pmax ? pmax : p);
    const long long per = 8ll * (6ll * nc) * (6ll * nc), cap = per > 0 ? (256ll << 20) / per : 256;
    if (cap < p) p = cap < 1 ? 1 : (int)cap;
    return p;
}
static std::atomic<int> g_ba_force_valu{0};  // PROCESS-WIDE test hook (include/velocity_hip.h)
extern "C" VH_API void vh_debug_ba_force_valu(int on) { g_ba_force_valu.store(on, std::memory_order_relaxed); }
extern "C" VH_API void vh_debug_pyr_rows(int rows) { vh_pyr_force_rows(rows == 2 || rows == 4 || rows == 8 ? rows : 0); }

extern "C" VH_API size_t vh_nls_batch_workspace(int nt, int nc) { return vh_ba_workspace_bytes(nt, nc, ba_parts(nt, nc)); }

extern "C" VH_API int vh_nls_batch(vh_ctx* c, const double* K_host, const double* z, double* x, int nt, int nc, int max_iter, double* trace,
                                   int* info, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!c || !K_host || nt < 1 || nc < 1 || max_iter < 1) return vh_fail(-1, "vh_nls_batch: bad arguments");
    if (nc > 255) return vh_fail(-1, "vh_nls_batch: at most 255 free cameras");
    if (workspace_bytes < vh_ba_workspace_bytes(nt, nc, ba_parts(nt, nc))) return vh_fail(-1, "vh_nls_batch: workspace too small");
    BaProblem P;
    P.graph_cache = c->ba_graph_on ? &c->ba_graphs : nullptr;
    P.ctx = c;
    for (int k = 0; k < 9; k++) P.K[k] = (double)K_host[k];
    P.z = z; P.x = x; P.trace = trace; P.info = info; P.workspace = workspace; P.nt = nt; P.nc = nc; P.max_iter = max_iter; P.nparts = ba_parts(nt, nc); P.force_valu = g_ba_force_valu.load(std::memory_order_relaxed);
    P.phase = -1; P.it = 0; P.add_identity = 1; P.count_cams = 1; P.defer_finalize = 0; P.model = 0;
    P.nwin = 1; P.ws_stride = P.z_stride = P.x_stride = P.trace_stride = P.info_stride = 0;
    P.nx_total = 3.0 * nt + 6.0 * nc; P.nz_total = 2.0 * nt * (nc + 1);
    int r = vh_ba_run(P, vh_ctx_bind(c, stream));
    if (r) return vh_fail(r, "vh_ba_run failed");
    return 0;
}

// nwin independent windows of the same shape through ONE launch sequence (grid.y = window): a sliding-window tracker has one window per
// video stream, and a single C5 window leaves the chip idle (its solve is one workgroup, its MFMA contraction is latency bound)
extern "C" VH_API int vh_nls_batch_multi(vh_ctx* c, const double* K_host, const double* z, double* x, int nt, int nc, int nwin, int max_iter,
                                         double* trace, int* info, void* workspace, size_t workspace_bytes_per_window, void* stream)
{
    if (!c || !K_host || nt < 1 || nc < 1 || max_iter < 1 || nwin < 1 || nwin > 65535) return vh_fail(-1, "vh_nls_batch_multi: bad arguments");
    if (nc > 255) return vh_fail(-1, "vh_nls_batch_multi: at most 255 free cameras");
    if (workspace_bytes_per_window < vh_ba_workspace_bytes(nt, nc, ba_parts(nt, nc)) || workspace_bytes_per_window % 256)
        return vh_fail(-1, "vh_nls_batch_multi: per-window workspace too small or not a multiple of 256 bytes");
    BaProblem P;
    P.graph_cache = c->ba_graph_on ? &c->ba_graphs : nullptr;
    P.ctx = c;
    for (int k = 0; k < 9; k++) P.K[k] = (double)K_host[k];
    P.z = z; P.x = x; P.trace = trace; P.info = info; P.workspace = workspace; P.nt = nt; P.nc = nc; P.max_iter = max_iter; P.force_valu = g_ba_force_valu.load(std::memory_order_relaxed);
    // fewer partial systems per window when many windows fill the chip anyway (the partials are reduced through HBM)
    int parts = ba_parts(nt, nc), cap = 512 / nwin < 16 ? 16 : 512 / nwin;
    P.nparts = nwin > 1 && parts > cap ? cap : parts;
    P.phase = -1; P.it = 0; P.add_identity = 1; P.count_cams = 1; P.defer_finalize = 0; P.model = 0;
    P.nx_total = 3.0 * nt + 6.0 * nc; P.nz_total = 2.0 * nt * (nc + 1);
    P.nwin = nwin; P.ws_stride = workspace_bytes_per_window; P.z_stride = (size_t)2 * nt * (nc + 1); P.x_stride = (size_t)3 * nt + 6 * (size_t)nc;
    P.trace_stride = (size_t)2 * max_iter; P.info_stride = 2;
    int r = vh_ba_run(P, vh_ctx_bind(c, stream));
    if (r) return vh_fail(r, "vh_ba_run failed");
    return 0;
}

// fcnNLS_batch2 (NLS.py:253-328): tie points + ONE joint rotation + a straight-line camera trajectory (el, az, one range per camera)
extern "C" VH_API int vh_nls_batch2(vh_ctx* c, const double* K_host, const double* z, double* x, int nt, int nc, int max_iter, double* trace,
                                    int* info, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!c || !K_host || nt < 1 || nc < 1 || max_iter < 1) return vh_fail(-1, "vh_nls_batch2: bad arguments");
    if (nc > 255) return vh_fail(-1, "vh_nls_batch2: at most 255 free cameras");
    if (workspace_bytes < vh_ba_workspace_bytes(nt, nc, ba_parts(nt, nc))) return vh_fail(-1, "vh_nls_batch2: workspace too small");
    BaProblem P;
    P.graph_cache = c->ba_graph_on ? &c->ba_graphs : nullptr;
    P.ctx = c;
    for (int k = 0; k < 9; k++) P.K[k] = (double)K_host[k];
    P.z = z; P.x = x; P.trace = trace; P.info = info; P.workspace = workspace; P.nt = nt; P.nc = nc; P.max_iter = max_iter; P.nparts = ba_parts(nt, nc); P.force_valu = 1;
    P.phase = -1; P.it = 0; P.add_identity = 1; P.count_cams = 1; P.defer_finalize = 0; P.model = 1;
    P.nwin = 1; P.ws_stride = P.z_stride = P.x_stride = P.trace_stride = P.info_stride = 0;
    P.nx_total = 3.0 * nt + nc + 5.0; P.nz_total = 2.0 * nt * (nc + 1);
    int r = vh_ba_run(P, vh_ctx_bind(c, stream));
    if (r) return vh_fail(r, "vh_ba_run failed");
    return 0;
}

// One phase of a point-sharded BA iteration: this rank owns nt tie points, all nc cameras are replicated.
// phase 0: init; 1: local normal equations -> [S | rhs | acc] span (the caller all-reduces it); 2: solve + update (the caller
// all-reduces acc again); 3: iteration record.  rank0 != 0 on exactly one rank (it adds the +I damping and counts the
// camera part of rms(delta)).  nt_total: points over all ranks.  span_offset / span_doubles (host, may be NULL) return
// where the all-reduce span lives inside the workspace.
extern "C" VH_API int vh_nls_batch_phase(vh_ctx* c, const double* K_host, const double* z, double* x, int nt, int nc, int nt_total, int rank0,
                                         int phase, int it, double* trace, int* info, void* workspace, size_t workspace_bytes,
                                         size_t* span_offset, size_t* span_doubles, void* stream)
{
    if (!c || !K_host || nt < 1 || nc < 1 || nt_total < nt) return vh_fail(-1, "vh_nls_batch_phase: bad arguments");
    if (nc > 255) return vh_fail(-1, "vh_nls_batch_phase: at most 255 free cameras");
    if (workspace_bytes < vh_ba_workspace_bytes(nt, nc, ba_parts(nt, nc))) return vh_fail(-1, "vh_nls_batch_phase: workspace too small");
    BaProblem P;
    P.graph_cache = c->ba_graph_on ? &c->ba_graphs : nullptr;
    P.ctx = c;
    for (int k = 0; k < 9; k++) P.K[k] = (double)K_host[k];
    P.z = z; P.x = x; P.trace = trace; P.info = info; P.workspace = workspace; P.nt = nt; P.nc = nc; P.max_iter = 1; P.nparts = ba_parts(nt, nc);
    P.force_valu = g_ba_force_valu.load(std::memory_order_relaxed);
    P.phase = phase; P.it = it; P.add_identity = rank0 ? 1 : 0; P.count_cams = rank0 ? 1 : 0; P.defer_finalize = 1; P.model = 0;
    P.nwin = 1; P.ws_stride = P.z_stride = P.x_stride = P.trace_stride = P.info_stride = 0;
    P.nx_total = 3.0 * nt_total + 6.0 * nc; P.nz_total = 2.0 * nt_total * (nc + 1);
    if (span_offset && span_doubles) vh_ba_exchange_span(P, span_offset, span_doubles);
    if (phase < 0 || phase > 3) return vh_fail(-1, "vh_nls_batch_phase: phase must be 0..3");
    int r = vh_ba_run(P, vh_ctx_bind(c, stream));
    if (r) return vh_fail(r, "vh_ba_run failed");
    return 0;
}
