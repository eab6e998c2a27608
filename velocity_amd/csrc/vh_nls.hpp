// NLS / MSV job descriptors and launchers (internal).
#pragma once
#include "vh_common.hpp"

// estimateWorldCameraPose (NLS.py:9-33) for one problem
struct PoseJob {
    double K[9];          // intrinsics, MATLAB row-vector layout (float32 values widened)
    double R[9];          // rotation used for the projection when mode == 0 (NLS_t ignores it, NLS.py:117)
    double x0[6];         // [rpy, t] start (NLS.py:20); mode 0 uses x0[3:6]
    const float* p;       // image points, rows selected by p_sel
    const double* pw;     // world points, rows selected by pw_sel
    const int* p_sel;     // may be null (identity)
    const int* pw_sel;    // may be null
    const int* n_ptr;     // device count (null -> n)
    int n;
    int mode;             // 0: fcnNLS_t, 1: fcnNLS_Rt
    float* t_out;         // 3
    double* R_out;        // 9 (may be null)
    double* res_out;      // 1   rms(p - p_proj)
    double* p_proj;       // n x 2 (may be null)
    int* info_out;        // 2   iterations, converged
};

// fcnMSV1_t (MSV.py:8-49)
struct MsvJob {
    double K[9];
    const float* P;       // float32 history (vidExample.py:128): entry (row, track, frame) at P[row * P_rs + track * P_ts + frame * P_fs]
    size_t P_rs, P_ts, P_fs;  // the reference's [5, N0, nhist]: (N0 nhist, nhist, 1); the session's frame-major [nhist, 5, N0]: (N0, 1, 5 N0)
    const float* B;       // [nhist, 14] float32
    const int* ids;       // ng global track ids (nonzero(vg)), may be null (identity)
    const int* ng_ptr;
    int ng, N0, nhist, nf;
    int max_iter;         // 1000 (MSV.py:24)
    int f32_rays;         // K and P are float32 -> pixel2uvec runs in float32 like numpy does
    double* U;            // scratch/out [3, nf, ng]
    double* b0;           // out [ng, 3]
    float* x_out;         // out 3
    int* info_out;        // out 2
};

void vh_launch_pose(const void* tab, size_t stride, int batch, int mode, int max_n, hipStream_t s);
void vh_launch_world2image(const double* C, const double* pw, int n, double* out, hipStream_t s);
void vh_launch_image2world(const double* Hi, const double* p, int n, double* out, hipStream_t s);
void vh_launch_pixel2uvec(double cx, double cy, double f, const double* p, int n, double* out, hipStream_t s);
void vh_launch_pixel2uvec_f32(float cx, float cy, float f, const float* p, int n, float* out, hipStream_t s);
void vh_launch_two_view(const double* A, const double* U, int nf, int nv, double* out, hipStream_t s);
void vh_launch_n_view(const double* A, const double* U, int nf, int nv, double* out, hipStream_t s);
void vh_launch_msv1(const MsvJob& job, hipStream_t s);
// the same for `batch` jobs that live `stride` bytes apart in device memory; a job runs when the int `frame_off` bytes from it equals fire_frame
void vh_launch_msv1_tab(const void* tab, size_t stride, ptrdiff_t frame_off, int fire_frame, int nf, int batch, hipStream_t s);
