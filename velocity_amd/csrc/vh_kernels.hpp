// Internal kernel-launcher interface of libvelocity_hip.
#pragma once
#include "vh_common.hpp"

// ROI warp stage of KLTregional (KLT.py:65-73)
struct WarpJob {
    ImgDesc src;      // current frame
    uint8_t* dst;     // ROI-sized output
    int dst_stride;
    int mode;         // -1: skip, 0: integer-shifted crop (zero padded), 1: affine remap INTER_LINEAR
    int x0, x1, y0, y1;
    int dx, dy;
    float T[6];       // 3x2 row-major float32
};

// cv2.estimateAffine2D stand-in (KLT.py:116,127): deterministic RANSAC + least-squares refit
#define VH_RANSAC_ITERS 2000
struct RansacJob {
    const float* from;    // n x 2 (all points; `valid` selects the m compacted pairs)
    const float* to;      // n x 2
    uint8_t* valid;       // n   in: pairs that take part; out (gate_valid=1): valid &= inlier
    const int* n_ptr;     // device count (null -> n)
    int n;
    int min_valid;        // run only when more than this many valid pairs (KLT.py:126: v.sum() > 10 -> 10)
    int gate_valid;       // 1: write the inlier mask back into `valid` (KLT.py:117)
    int* idx;             // scratch n    : compacted index list
    float4* pairs;        // scratch n    : compacted pairs (from.x, from.y, to.x, to.y): one coalesced 16-byte load per pair when scoring
    int* counts;          // scratch VH_RANSAC_ITERS : inliers per hypothesis
    int* m_out;           // scratch 1    : number of valid pairs
    int* bound;           // scratch 1    : upper bound of the hypotheses the sequential rule can still reach
    double* M;            // out 6        : 2x3 row-major affine
    uint8_t* inl;         // out n        : inlier mask over ALL points (0 where !valid)
    int* status;          // out 1        : 1 = model found
};

// pyramid build request: level l+1 of *pyr is computed from level l when enabled
struct PyrBuild {
    PyrDesc* pyr;
    int enable;
    int pad;
};

void vh_launch_resize_quarter(const void* src_tab, const void* dst_tab, size_t tab_stride, int per_stream, int batch, int max_dw, int max_dh,
                              hipStream_t s);
void vh_launch_pyr_down_ws(const void* pb_tab, size_t ws_stride, int batch, int lvl, int max_w0, int max_h0, hipStream_t s);
void vh_launch_resize_nearest(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, int dw, int dh, size_t dstride, double ifx, double ify,
                              hipStream_t s);
void vh_launch_bgr2gray(const uint8_t* bgr, int w, int h, size_t sstride, uint8_t* gray, size_t dstride, hipStream_t s);
// fused frame ingest (BGR -> gray + quarter-scale image), one descriptor per frame
struct IngestJob {
    const uint8_t* bgr;   // h x w x 3, rows bgr_stride bytes apart (null: skip)
    uint8_t* gray;        // h x w, rows gray_stride bytes apart
    uint8_t* small;       // round(h/4) x round(w/4), rows small_stride bytes apart (may be null)
    int w, h, bgr_stride, gray_stride, dw, dh, small_stride;
};
void vh_launch_ingest_bgr(const IngestJob* jobs_dev, int count, int max_w, int max_h, hipStream_t s);
void vh_pyr_force_rows(int rb);
void vh_launch_roi_warp(const void* job_tab, size_t tab_stride, int batch, int max_w, int max_h, hipStream_t s);
int vh_launch_lk(const void* job_tab, size_t tab_stride, int batch, int max_n, int win, hipStream_t s, int* route_out = nullptr, int* tpw_out = nullptr);
int vh_lk_route(int batch, int max_n, int win);           // the kernel such a launch takes (ids of vh_debug_force_generic_lk)
const char* vh_lk_route_name(int route, int win);
struct StreamWS;
// returns true when the glue that follows the call in KLTmain (glue = 1: stage 1 -> 2, 2: stage 2 -> 3; glue_ws = the streams' workspaces) ran as the
// epilogue of the fused kernel, false when the caller has to launch it (no glue asked for, or the three-kernel path was taken)
bool vh_launch_ransac(const void* job_tab, size_t tab_stride, int batch, int max_n, hipStream_t s, StreamWS* glue_ws = nullptr, int glue = 0);
