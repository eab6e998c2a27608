// Per-stream device workspace shared by the KLTmain pipeline (vh_api.hip) and the session (vh_session.hip).
#pragma once
#include "../../include/velocity_hip.h"
#include "vh_kernels.hpp"
#include "vh_nls.hpp"

// ---------------------------------------------------------------------------------------------------------------
// per-stream device workspace
// ---------------------------------------------------------------------------------------------------------------
struct StreamBufs {  // fixed after vh_ctx_create
    uint8_t* small0[2];                    // quarter-scale frames (ping-pong) when the caller passes none
    uint8_t* small_lv[2][VH_MAX_LEVELS];   // quarter-scale pyramid levels >= 1 (ping-pong)
    uint8_t* roi_lv[2][VH_MAX_LEVELS];     // ROI pyramid levels >= 1 of the previous (0) / current (1) frame
    uint8_t* warp;                         // shifted crop (stage 2) / affine-warped ROI (stage 3)
    float* p_small;
    float* p_coarse;
    uint8_t* v_small;
    uint8_t* v_coarse;
    uint8_t* v_all;                        // all-ones mask for the stateless RANSAC entry
    uint8_t* inl;
    int* idx;
    float4* pairs;
    int* counts;
    int* order;                            // launch order of the tracks (k_klt_setup's spatial counting sort), max_pts entries
};

struct KltIO {  // one KLTmain call (KLT.py:99)
    const uint8_t* im;
    const uint8_t* im0;
    const uint8_t* im0_small;  // may be null
    const float* p0;
    const int* n_ptr;          // device count (null -> n)
    float* p_all;
    uint8_t* v;
    uint8_t* im_small;         // may be null (internal buffer)
    int* flags;                // may be null
    int w, h, stride, stride0, n;
    int reuse_prev_small;      // 1: small_lv[1 - pp] already holds the pyramid of im0_small (session mode)
    int have_small;            // 1: im_small already holds the quarter-scale image of `im` (fused ingest): no resize
    vh_lk_params coarse, fine;
    float fbt_coarse, fbt_fine;  // 1.0, 0.3 (KLT.py:124,133)
};

struct StreamWS {
    LKJob lk;
    WarpJob warp;
    RansacJob ransac;
    PyrBuild pb[2];
    ImgDesc rs_src[2], rs_dst[2];  // quarter-scale resize table: [0] current frame, [1] previous frame
    KltIO io;
    StreamBufs bufs;
    double M[6];
    double t_trans[2];
    int roi[4];
    int dxy[2];
    // per KLTmain stage: Newton iterations ([slot][0]) and template set-ups ([slot][1]), summed over VH_LK_STAT_SLOTS slots of one 128-byte line each.  A
    // workgroup adds to slot (its launch slot mod VH_LK_STAT_SLOTS): with ONE pair of counters per stream the 2 x 2000 same-address atomics of a single-stream
    // launch serialised in the L2 -- ~45 us of every ~62 us LK launch, 0.33 -> 0.21 ms per single-stream frame step once they were spread out
    unsigned long long lk_stats[3][VH_LK_STAT_SLOTS][16];
    int n, m, rstatus, flags, pp, rbound;
    const int* order;  // bufs.order when this call launches its LK kernels in spatial order, else null
};

// scratch of the frame-0 detector (vh_init.hip: goodFeaturesToTrack / cornerSubPix), owned by the context that uses it
struct InitScratch {
    int* dxy;
    float* resp;
    unsigned long long *keys, *sorted;
    unsigned* counters;  // [0] max (ordered bits), [1] candidate count, [2..3] pose info, [4] corner count of vh_frame0_init
    float* mask;         // cornerSubPix Gaussian windows of every half-size 1..7, back to back
    void* sort_tmp;
    size_t sort_bytes, pixels;
};

struct vh_ctx {
    int batch, max_w, max_h, max_pts, sw, sh;
    char* arena;
    size_t arena_bytes;
    StreamWS* d_ws;
    StreamBufs* h_bufs;  // host copy of every stream's buffer table
    // optional per-stage HIP-event timing of the LK launches (bench.py roofline leg)
    int prof_on, prof_n, prof_cap;  // prof_on: 0 off, 1 LK launches only, 2 every stage
    int prof_dropped;                // launches that found the record table full since vh_profile_begin (vh_profile_end* report them as an error)
    int prof_light;                  // vh_profile_detail(ctx, 0): the next vh_profile_begin times the LK launches only
    hipEvent_t* prof_ev;  // 2 * prof_cap events: start/stop pairs
    int* prof_stage;
    double* d_small;     // 64 doubles of scratch for host-provided small matrices
    void* ba_graphs;     // replayable launch sequences of whole BA solves (vh_ba.hip), created on demand
    int ba_graph_on;     // vh_ba_graph_replay: the caller promises pointer-stable buffers (default 0: every solve is launched plainly)
    hipStream_t bound_stream;  // the stream whose work may still read this context's job descriptors (vh_ctx_bind)
    hipEvent_t bound_ev;       // recorded on bound_stream by vh_ctx_release at the end of every entry point: what a rebind waits for
    int bound;                 // 0: never used, 1: bound
    int lk_tpw[3];             // launch slots per workgroup of those launches (1 unless the one-wavefront LDS-staged kernel looped; vh_profile_lk_tpw)
    int lk_route[3];           // kernel route (vh_lk_route ids) the last KLTmain took for its three LK launches (vh_profile_lk_routes)
    int lk_win[3];
    InitScratch init;          // created by the first frame-0 call (vh_init.hip)
};

// A vh_ctx parks the job descriptors of the calls in flight, so it serves ONE HIP stream at a time.  Enforced on the DEVICE: every entry point ends
// with vh_ctx_release (an event record on its stream); an entry point that arrives with another stream makes that stream wait for the event
// (hipStreamWaitEvent) before it rebinds.  No host blocking, legal under stream capture, and the previous stream handle is never touched again (it
// may have been destroyed: only the event is used).
static inline hipStream_t vh_ctx_bind_raw(vh_ctx* c, void* stream, int* err = nullptr)
{
    hipStream_t s = (hipStream_t)stream;
    if (err) *err = 0;
    if (c) {
        if (c->bound && c->bound_stream != s && c->bound_ev) {
            // A capturing stream must not wait for an event recorded OUTSIDE its capture (an isolation violation that can invalidate the caller's
            // capture), so there is nothing to wait with there.  The rule is the caller's -- a context does not change stream inside a capture (the
            // work it queued on its previous stream must have been ordered before the capture began; include/velocity_hip.h "Conventions") -- and it is
            // CHECKED: the rebind is accepted only when the earlier work has provably completed (hipEventQuery); otherwise the entry point fails
            // instead of letting the captured launches overwrite job descriptors the previous stream may still be reading.
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
            if (cs == hipStreamCaptureStatusNone) {
                if (hipStreamWaitEvent(s, c->bound_ev, 0) != hipSuccess) (void)hipGetLastError();
            } else {
                // (an event query counts as a "potentially unsafe" call while a capture in global / thread-local mode is under way: relaxed for the query only)
                hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
                const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
                const hipError_t q = hipEventQuery(c->bound_ev);
                if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
                if (q != hipSuccess) {
                    (void)hipGetLastError();
                    if (err) *err = 1;
                    return s;  // (the context stays bound to its previous stream)
                }
            }
        }
        c->bound_stream = s;
        c->bound = 1;
    }
    return s;
}
// end of an entry point: marks how far the bound stream has got with this context's descriptors (one event record, no synchronisation).  It is recorded
// by every call because a later rebind may find the previous stream already destroyed: only the event may be touched then.
static inline void vh_ctx_release(vh_ctx* c)
{
    if (!c || !c->bound) return;
    if (!c->bound_ev && hipEventCreateWithFlags(&c->bound_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->bound_ev = nullptr; return; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->bound_stream, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs != hipStreamCaptureStatusNone) return;  // inside a capture the launches are graph nodes: ordering is the graph's
    if (hipEventRecord(c->bound_ev, c->bound_stream) != hipSuccess) (void)hipGetLastError();
}


// scope of one entry point: binds on construction, releases (event record) when the entry point returns, whichever return path it takes
struct vh_ctx_bind {
    vh_ctx* c;
    int err;  // 1: the context was asked to change stream inside a stream capture while its earlier work is still running (see vh_ctx_bind_raw)
    hipStream_t s;
    vh_ctx_bind(vh_ctx* ctx, void* stream) : c(ctx), err(0), s(vh_ctx_bind_raw(ctx, stream, &err)) {}
    ~vh_ctx_bind() { if (!err) vh_ctx_release(c); }
    vh_ctx_bind(const vh_ctx_bind&) = delete;
    vh_ctx_bind& operator=(const vh_ctx_bind&) = delete;
    operator hipStream_t() const { return s; }
};


#define VH_BIND(ctx_, stream_)                                                                                                                              \
    vh_ctx_bind bound_(ctx_, stream_);                                                                                                                     \
    if (bound_.err)                                                                                                                                        \
        return vh_fail(-6, "the context was handed another stream inside a stream capture while work it queued on its previous stream is still running: " \
                           "order that work before the capture begins (include/velocity_hip.h, Conventions)")

// tracker-session state of one video stream (vh_session.hip); declared here because the KLTmain set-up kernel (vh_api.hip) fetches a
// frame's inputs straight from it (one launch less per frame than a separate "prepare" kernel)
struct SessStream {  // device resident, one per video stream
    // track state (vidExample.py:125-129)
    uint8_t* vg;     // N0  global validity of the initial tracks
    uint8_t* vp;     // N0  tracks used for the pose fit
    float* p_cur;    // n_cur x 2 compacted current points
    int* ids;        // n_cur global ids of the compacted rows (= nonzero(vg))
    double* p3;      // N0 x 3 world points
    float* P;        // [5, N0, nhist] history, NaN padded
    float* B;        // [nhist, 14]
    float* S;        // [nhist, 9]
    // per-frame scratch
    float* p_all;    // KLTmain output before compaction
    uint8_t* v;      // KLTmain status
    int* sel_p;      // pose rows of p_cur      (p[vp[vg]], vidExample.py:139)
    int* sel_pw;     // pose rows of p3         (p3[vp])
    double* p_proj;  // n_pose x 2
    double* msv_U;   // 3*16*N0 scratch of fcnMSV1_t
    double* msv_b0;  // N0 x 3
    uint8_t* small[2];
    const uint8_t* im0;
    PoseJob pose;
    MsvJob msv;      // this stream's fcnMSV1_t job (vidExample.py:155-158), fixed at session creation
    double K[9];
    double res;
    float t[3];
    float msv_x[3];
    float r_total, t0;
    int pose_info[2], msv_info[2];
    int N0, nhist, n_cur, n_pose, frame_i, pp, klt_flags, w, h, stride;
    int small_ready;  // frame index whose quarter-scale image vh_session_ingest_bgr has already written into small[pp] (0: none)
};


// runs KLTmain (KLT.py:99-134) for streams [slot, slot+count) whose KltIO has been written (host or device side)
int vh_run_klt_main(vh_ctx* c, int slot, int count, hipStream_t s, const vh_lk_params& coarse, const vh_lk_params& fine,
                    const SessStream* sess = nullptr, const uint8_t* const* frames = nullptr, int n_max = 0);
int vh_fail(int code, const char* msg);
void vh_init_scratch_free(vh_ctx* c);

// optional HIP-event timing of individual launches (vh_profile_begin / vh_profile_end_stages): stage ids
enum { VH_PROF_LK0 = 0, VH_PROF_LK1 = 1, VH_PROF_LK2 = 2, VH_PROF_WARP = 3, VH_PROF_PYR = 4, VH_PROF_RANSAC = 5, VH_PROF_RESIZE = 6, VH_PROF_SESSION = 7,
       VH_PROF_BA_JAC = 8, VH_PROF_BA_SCHUR = 9, VH_PROF_BA_REDUCE = 10, VH_PROF_BA_SOLVE = 11, VH_PROF_BA_UPDATE = 12, VH_PROF_STAGES = 16 };
// start of a profiled launch: returns the record index (or -1 when profiling is off / the record table is full); vh_prof_stop closes it
static inline int vh_prof_start(vh_ctx* c, hipStream_t s, int level = 2)
{
    // level 1: the three LK launches of a frame step (always timed while profiling is on); level 2: every other stage (vh_profile_detail)
    if (!c || c->prof_on < level) return -1;
    if (c->prof_n >= c->prof_cap) {  // table sized for fewer launches than were issued: counted, and vh_profile_end / _end_stages fail loudly
        c->prof_dropped++;
        return -1;
    }
    (void)hipEventRecord(c->prof_ev[2 * c->prof_n], s);
    return c->prof_n;
}
static inline void vh_prof_stop(vh_ctx* c, int rec, int stage, hipStream_t s)
{
    if (rec < 0) return;
    (void)hipEventRecord(c->prof_ev[2 * rec + 1], s);
    c->prof_stage[rec] = stage;
    c->prof_n = rec + 1;
}
