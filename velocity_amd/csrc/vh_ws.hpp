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
    unsigned long long lk_stats[3][2];  // per KLTmain stage: Newton iterations, template set-ups (profiling aid)
    int n, m, rstatus, flags, pp, rbound;
};

struct vh_ctx {
    int batch, max_w, max_h, max_pts, sw, sh;
    char* arena;
    size_t arena_bytes;
    StreamWS* d_ws;
    StreamBufs* h_bufs;  // host copy of every stream's buffer table
    // optional per-stage HIP-event timing of the LK launches (bench.py roofline leg)
    int prof_on, prof_n, prof_cap;
    hipEvent_t* prof_ev;  // 2 * prof_cap events: start/stop pairs
    int* prof_stage;
    double* d_small;     // 64 doubles of scratch for host-provided small matrices
};


// runs KLTmain (KLT.py:99-134) for streams [slot, slot+count) whose KltIO has been written (host or device side)
int vh_run_klt_main(vh_ctx* c, int slot, int count, hipStream_t s, const vh_lk_params& coarse, const vh_lk_params& fine);
int vh_fail(int code, const char* msg);
