/*
 * libvelocity_hip -- C ABI of the MI355X-native KLT + NLS hot path of ultralytics/velocity.
 *
 * The reference has no FFI / plugin interface: its boundary is the set of Python call sites in vidExample.py
 * (SURVEY.md section 8b).  Every entry point below names the reference function (file:line) it replaces; the Python
 * binding a maintainer would add is velocity_amd/_lib.py (ctypes) and is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is marked "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on it, inputs are
 *     borrowed and never modified, outputs are caller-allocated;
 *   - images are uint8, pixel (x,y) at base[y*stride + x]; points are float32 (x,y) pairs; poses follow the
 *     reference's row-vector layout (uv1 ~ [X Y Z] @ K, X_cam = X_w @ R + t, affine [x y 1] @ T with T 3x2);
 *   - a vh_ctx serves ONE HIP stream at a time: the stateless entry points (vh_pyr_lk, vh_pose, vh_remap_affine, ...) park their small job
 *     descriptors in slot 0 of the workspace they are given, so calls that may overlap on different streams / threads need their own vh_ctx
 *     (both bindings do this: velocity_amd/_lib.py::workspace and vh_torch_ops.cpp keep one per (device, stream)).  The library enforces the rule:
 *     an entry point that is handed another stream than the context's previous call first waits for the work queued through the context on that
 *     previous stream (a serialisation, never a race);
 *   - camera intrinsics K_host are 9 float64 in the reference's MATLAB layout [[fx,0,0],[s,fy,0],[cx,cy,1]] (a float32 K widens exactly);
 *   - return value 0 = success, otherwise a hipError_t (or a negative vh error); vh_last_error() describes it.
 *     Numerical non-convergence is NOT an error: like the reference (NLS.py:126-127,178-179; KLT.py:129) the call
 *     succeeds and reports it through an info/flags output so the host shim can print the same warning.
 */
#ifndef VELOCITY_HIP_H
#define VELOCITY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VH_API __attribute__((visibility("default")))

typedef struct vh_ctx vh_ctx;         /* device workspace for `batch` independent video streams               */
typedef struct vh_session vh_session; /* per-frame tracker state of `batch` streams (vidExample.py loop body) */

/* Lucas-Kanade parameters: cv2 winSize=(win,win), maxLevel, criteria=(EPS|COUNT, max_count, eps)  (KLT.py:106-107) */
typedef struct {
    int win;
    int max_level;
    int max_count;
    double eps;
} vh_lk_params;

/* device pointers to the intermediate results of the last vh_klt_main call of one stream (parity tests) */
typedef struct {
    const float* p_small;     /* n x 2  stage-1 result at full resolution (KLT.py:114-115)  */
    const uint8_t* v_small;   /* n      after the RANSAC inlier gate (KLT.py:117)           */
    const double* t_trans;    /* 2      mean translation (KLT.py:121-123)                   */
    const int* roi;           /* 4      x0,x1,y0,y1 (KLT.py:60)                             */
    const float* p_coarse;    /* n x 2  stage-2 result (KLT.py:124)                         */
    const uint8_t* v_coarse;  /* n                                                          */
    const double* t23;        /* 6      2x3 affine (KLT.py:127)                             */
    const uint8_t* warped;    /* ROI-sized affine warp, row stride = roi width rounded up to a multiple of 4 (KLT.py:73)  */
    const int* flags;         /* 1      bit0: coarse-affine failure (KLT.py:128-130)        */
} vh_klt_stages;

VH_API int vh_version(void);
/* identity of the sources + compiler this binary was built from: "<sha256(csrc, this header, flags)[:24]>-<sha256(hipcc --version)[:8]>"
 * (velocity_amd/_build.py::build_id).  The Python loader refuses a library whose id is not the tree's; bench.py and pytest print it. */
VH_API const char* vh_build_id(void);
VH_API const char* vh_last_error(void);
/* utility: synchronous device -> host copy of a raw device pointer (used to read the stage pointers below) */
VH_API int vh_copy_to_host(void* dst_host, const void* src_dev, size_t bytes, void* stream);

/* ---- workspace ------------------------------------------------------------------------------------------------ */
VH_API int vh_ctx_create(vh_ctx** out, int batch, int max_w, int max_h, int max_pts);
VH_API void vh_ctx_destroy(vh_ctx* ctx);

/* ---- image stages (K1, K2, K8, K9) ---------------------------------------------------------------------------- */
/* cv2.resize(im, (0,0), fx=.25, fy=.25, INTER_NEAREST), utils/KLT.py:111-113.  dst: round(h/4) x round(w/4), dense */
VH_API int vh_resize_quarter(vh_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, void* stream);
/* frame ingest, optional rescale: cv2.resize(im, (0,0), fx=scale, fy=scale, interpolation=INTER_NEAREST), vidExample.py:99-102.
 * dst: round(h fy) x round(w fx) with row stride dst_stride */
VH_API int vh_resize_nearest(vh_ctx* ctx, const uint8_t* src, int w, int h, int stride, double fx, double fy, uint8_t* dst, int dst_stride,
                             void* stream);
/* frame ingest, cv2.cvtColor(imbgr, cv2.COLOR_BGR2GRAY), vidExample.py:91.
 * bgr: uint8 [h][w][3] with row stride stride_bytes; gray: uint8 [h][w] with row stride gray_stride */
VH_API int vh_bgr2gray(vh_ctx* ctx, const uint8_t* bgr, int w, int h, int stride_bytes, uint8_t* gray, int gray_stride, void* stream);
/* the same pass also writing the quarter-scale image KLTmain starts from (cv2.resize(.25, INTER_NEAREST), KLT.py:111-113): small = round(h/4) x round(w/4),
 * dense (may be NULL: gray only) */
VH_API int vh_ingest_bgr(vh_ctx* ctx, const uint8_t* bgr, int w, int h, int stride_bytes, uint8_t* gray, int gray_stride, uint8_t* small, void* stream);
/* cv2.pyrDown as used inside cv2.calcOpticalFlowPyrLK (KLT.py:45,48).  dst: (h+1)/2 x (w+1)/2, dense */
VH_API int vh_pyr_down(vh_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, void* stream);
/* meshgrid + affine + cv2.remap(INTER_LINEAR), utils/KLT.py:70-73.  T: host, 3x2 row-major float32.  dst dense ROI */
VH_API int vh_remap_affine(vh_ctx* ctx, const uint8_t* im, int w, int h, int stride, const float* T_host, int x0, int x1, int y0,
                           int y1, uint8_t* dst, void* stream);
/* im[y0+dy:y1+dy, x0+dx:x1+dx] with zero padding outside the frame, utils/KLT.py:65-68 (SURVEY App. B intent) */
VH_API int vh_crop_shift(vh_ctx* ctx, const uint8_t* im, int w, int h, int stride, int x0, int x1, int y0, int y1, int dx, int dy,
                         uint8_t* dst, void* stream);
/* boundingRect(x, imshape, border), utils/images.py:9-19.  roi_out: device int[4] = x0,x1,y0,y1 */
VH_API int vh_bounding_rect(vh_ctx* ctx, const float* p, int n, int imw, int imh, int bx, int by, int* roi_out, void* stream);

/* ---- tracker -------------------------------------------------------------------------------------------------- */
/* cv2calcOpticalFlowPyrLK(im1, im2, p1, None, fbt, **lk), utils/KLT.py:37-51.  fbt < 0 means fbt=None.
 * Outputs: p2 n x 2, v n (uint8 0/1), err n (may be NULL), fbe n (may be NULL). */
VH_API int vh_pyr_lk(vh_ctx* ctx, const uint8_t* im1, const uint8_t* im2, int w, int h, int stride1, int stride2, const float* p1,
                     int n, const vh_lk_params* lk_host, float fbt, float* p2, uint8_t* v, float* err, float* fbe, void* stream);
/* cv2.estimateAffine2D(from[valid], to[valid], method=RANSAC), utils/KLT.py:116,127 (deterministic stand-in, see
 * DESIGN.md).  valid may be NULL (all).  Outputs: M device double[6] (2x3), inl device uint8[n], status device int[1]. */
VH_API int vh_ransac_affine(vh_ctx* ctx, const float* from, const float* to, const uint8_t* valid, int n, double* M, uint8_t* inl,
                            int* status, void* stream);
/* KLTmain(im, im0, im0_small, p0), utils/KLT.py:99-134, for stream slot `slot` of the workspace.
 * im0_small may be NULL.  Outputs: p_all n x 2 (every point; the shim returns p_all[v]), v n, im_small
 * round(h/4) x round(w/4), flags device int[1] (bit0: "KLT coarse-affine failure", KLT.py:129). */
VH_API int vh_klt_main(vh_ctx* ctx, int slot, const uint8_t* im, const uint8_t* im0, const uint8_t* im0_small, int w, int h, int stride,
                       int stride0, const float* p0, int n, const vh_lk_params* coarse_host, const vh_lk_params* fine_host,
                       float* p_all, uint8_t* v, uint8_t* im_small, int* flags, void* stream);
VH_API int vh_klt_stage_ptrs(vh_ctx* ctx, int slot, vh_klt_stages* out_host);
/* KLTregional(im0, im, p0, T, lk, fbt, translateFlag), utils/KLT.py:55-95.  T_host: 3x2 row-major float32 (host).
 * Outputs: p_out n x 2 (frame coordinates), v_out n, roi_out device int[4] = x0,x1,y0,y1 (may be NULL). */
VH_API int vh_klt_regional(vh_ctx* ctx, const uint8_t* im0, const uint8_t* im, int w, int h, int stride0, int stride, const float* p0,
                           int n, const float* T_host, const vh_lk_params* lk_host, float fbt, int translate, float* p_out,
                           uint8_t* v_out, int* roi_out, void* stream);

/* ---- NLS pose (K11-K13) ---------------------------------------------------------------------------------------- */
/* estimateWorldCameraPose(K, p, p3, t, R, findR), utils/NLS.py:9-33 -> fcnNLS_t (:102-129) / fcnNLS_Rt (:133-183).
 * K_host: 9 doubles (MATLAB layout; a float32 K widens exactly -- estimateWorldCameraPose does K.astype(float), NLS.py:22-24),
 * x0_host: 6 doubles [rpy(R), t], R_host: 9 doubles.
 * Outputs: t_out float[3]; R_out double[9] (findR: float32-rounded rpy2dcm, else R); res_out double[1] = rms(p - p_proj);
 * p_proj double[n x 2] (may be NULL); info int[2] = {iterations, converged}. */
VH_API int vh_pose(vh_ctx* ctx, const double* K_host, const float* p, const double* pw, int n, const double* x0_host,
                   const double* R_host, int findR, float* t_out, double* R_out, double* res_out, double* p_proj, int* info,
                   void* stream);
/* world2image(K, R, t, pw), utils/common.py:58-64.  C_host = [R; t] @ K (4x3 row-major, host).  out n x 2 */
VH_API int vh_world2image(vh_ctx* ctx, const double* C_host, const double* pw, int n, double* out, void* stream);
/* image2world(K, R, t, p), utils/common.py:49-55.  Hi_host = inv([R[0:2]; t] @ K) (3x3, host).  out n x 2 */
VH_API int vh_image2world(vh_ctx* ctx, const double* Hi_host, const double* p, int n, double* out, void* stream);
/* pixel2uvec(K, p), utils/common.py:122-126.  out n x 3 */
VH_API int vh_pixel2uvec(vh_ctx* ctx, double cx, double cy, double f, const double* p, int n, double* out, void* stream);
/* the same when K and p are float32 on the caller's side: numpy then computes (and returns) float32.  p n x 2, out n x 3 float32 */
VH_API int vh_pixel2uvec_f32(vh_ctx* ctx, float cx, float cy, float f, const float* p, int n, float* out, void* stream);

/* ---- triangulation (K15, K16) ---------------------------------------------------------------------------------- */
/* fcn2vintercept(A, U), utils/MSV.py:98-142.  A [nf,3], U [3,nf,nv], out [nv,3] (all device, float64) */
VH_API int vh_two_view_intercept(vh_ctx* ctx, const double* A, const double* U, int nf, int nv, double* out, void* stream);
/* fcnNvintercept(A, U), utils/MSV.py:146-175: N-ray least-squares intersection.  Same layouts. */
VH_API int vh_n_view_intercept(vh_ctx* ctx, const double* A, const double* U, int nf, int nv, double* out, void* stream);
/* fcnMSV1_t(K, P, B, vg, ii), utils/MSV.py:8-49.  P float32 [5,N0,nhist], B float32 [nhist,14], ids = nonzero(vg)
 * (int32, ng entries).  f32_rays: K and P were float32 on the caller's side (numpy then builds the rays in float32).
 * Outputs: x_out float[3], b0 double[ng x 3], info int[2]; U_scratch double[3*(ii+1)*ng]. */
VH_API int vh_msv1_t(vh_ctx* ctx, const double* K_host, const float* P, const float* B, const int* ids, int ng, int N0, int nhist,
                     int ii, int f32_rays, double* U_scratch, float* x_out, double* b0, int* info, void* stream);


/* ---- bundle adjustment (K14) ------------------------------------------------------------------------------------- */
/* fcnNLS_batch(K, P, pw, cw), utils/NLS.py:186-250: dense LM over tie points and cameras 1..nc, +I damping, step 0.9.
 * The host shim applies the track filter and packs z / x exactly as NLS.py:190-203 does:
 *   z [2*nt*(nc+1)] float64 = [all u | all v], camera-major / track-minor;  x [3 nt + 6 nc] = points | cam pos | cam rpy.
 * x is updated in place.  trace [max_iter][2] = (rms(z - zhat), rms(delta)) per iteration (what NLS.py:238 prints),
 * info int[2] = {iterations, converged}.  workspace: vh_nls_batch_workspace(nt, nc) bytes of device memory.
 * nc <= 255 free cameras (the reference has no limit; -1 above).  The reduced camera system is FORMED on the matrix cores -- one 128-wide Schur launch up
 * to 21 cameras, a 256-wide two-pass Schur kernel for 22..42, a materialised Z + K-split SYRK for 43..255 -- and SOLVED by a matrix-core block
 * Gauss-Jordan up to 20 cameras (124 unknowns), a register-resident VALU Gauss-Jordan at 21 cameras (126 unknowns), and from 22 cameras by a
 * left-looking Cholesky, one launch per 32 columns.  The workspace grows with (6 nc)^2 per partial system: ask vh_nls_batch_workspace, never guess. */
VH_API size_t vh_nls_batch_workspace(int nt, int nc);
/* Opt-in (default OFF): replay a repeated whole solve as ONE hipGraph launch.  A captured sequence bakes the device pointers it was captured with
 * (z, x, trace, info, workspace), so the caller promises POINTER STABILITY: the same buffers are passed on every call (a sliding-window solver that
 * owns its arrays; bench.py).  With on = 1 the second call with an identical job descriptor captures (one synchronous instantiate in that call's
 * latency), later ones replay; any other descriptor is launched plainly; at most 8 sequences are kept (least recently used is dropped after a stream
 * synchronisation).  Callers whose buffers come from an allocator per call (the torch ops, the numpy shim) leave it off.  on = 0 also stops the
 * replay of sequences captured earlier. */
VH_API int vh_ba_graph_replay(vh_ctx* ctx, int on);
VH_API int vh_nls_batch(vh_ctx* ctx, const double* K_host, const double* z, double* x, int nt, int nc, int max_iter, double* trace,
                        int* info, void* workspace, size_t workspace_bytes, void* stream);

/* nwin independent fcnNLS_batch problems of the same shape (one sliding window per video stream) solved by ONE launch sequence
 * (grid.y = window): z [nwin][2 nt (nc+1)], x [nwin][3 nt + 6 nc], trace [nwin][max_iter][2], info [nwin][2]; workspace = nwin blocks
 * of workspace_bytes_per_window >= vh_nls_batch_workspace(nt, nc) bytes (a multiple of 256).  Every window takes exactly the steps
 * vh_nls_batch would take on it (its own stop flag included). */
VH_API int vh_nls_batch_multi(vh_ctx* ctx, const double* K_host, const double* z, double* x, int nt, int nc, int nwin, int max_iter,
                              double* trace, int* info, void* workspace, size_t workspace_bytes_per_window, void* stream);

/* fcnNLS_batch2(K, P, pw, cw), utils/NLS.py:253-328: the constrained sibling -- tie points, ONE joint rotation applied to the
 * points and a straight-line camera trajectory (elevation, azimuth, one range per camera 1..nc; camera 0 at the origin).
 * Same z packing, damping (+I), step (0.9) and stop rule (rms(delta) < 1e-7); the reference runs at most 20 iterations.
 *   x [3 nt + 5 + nc] float64 = points | joint rpy (3) | el, az | ranges (nc)   (NLS.py:274), updated in place.
 * trace / info / workspace as vh_nls_batch (the same workspace size serves both). */
VH_API int vh_nls_batch2(vh_ctx* ctx, const double* K_host, const double* z, double* x, int nt, int nc, int max_iter, double* trace,
                         int* info, void* workspace, size_t workspace_bytes, void* stream);

/* One phase of a point-sharded BA iteration (multi-GPU fcnNLS_batch): this rank owns nt of the nt_total
 * tie points, the nc free cameras are replicated.  phase 0: init; 1: local normal equations -> [S | rhs | sums] span inside
 * the workspace (the caller all-reduces span_doubles float64 at span_offset bytes); 2: solve + update (the caller
 * then all-reduces ONLY sum delta^2 = the span's third-from-last double; sum r^2 was final after phase 1); 3: iteration record (trace, info, stop flag).  rank0 != 0 on one rank. */
VH_API int vh_nls_batch_phase(vh_ctx* ctx, const double* K_host, const double* z, double* x, int nt, int nc, int nt_total, int rank0,
                              int phase, int it, double* trace, int* info, void* workspace, size_t workspace_bytes,
                              size_t* span_offset, size_t* span_doubles, void* stream);

/* ---- frame-0 initialisation ---------------------------------------------------------- */
/* cv2.goodFeaturesToTrack(roi, maxCorners, qualityLevel, 0, blockSize=block, useHarrisDetector=True, k), vidExample.py:110.
 * corners: device float [max_corners x 2] (x, y) sorted by response; count: device int[1] */
VH_API int vh_good_features(vh_ctx* ctx, const uint8_t* im, int w, int h, int stride, int max_corners, double quality, int block,
                            double k, float* corners, int* count, void* stream);
/* cv2.cornerSubPix(im, pts, (win,win), (-1,-1), (EPS+MAX_ITER, max_iter, eps)), vidExample.py:113-115.  pts refined in place */
VH_API int vh_corner_subpix(vh_ctx* ctx, const uint8_t* im, int w, int h, int stride, float* pts, int n, int win, int max_iter,
                            double eps, void* stream);

/* cv2.cornerSubPix / goodFeaturesToTrack scratch of this context (Harris planes, sort keys, Gaussian masks): created by the first frame-0 call, or
 * explicitly here -- e.g. before a stream capture, inside which it cannot be allocated.  Sized for max(w x h, the context's max_w x max_h). */
VH_API int vh_init_reserve(vh_ctx* ctx, int w, int h, void* stream);
/* Frame-0 initialisation of one video, vidExample.py:105-127, as ONE device-resident launch sequence (no host round trip between its steps):
 *   boxa / boxb = boundingRect(q, imshape, border=(0,0) / (border_x, border_y))                (:107-108, images.py:9-19; q_host: the 4 clicked corners, host)
 *   p = goodFeaturesToTrack(im[boxb], max_corners, quality, 0, blockSize=block, useHarrisDetector=True, k) + boxb origin      (:109-112)
 *   p = cornerSubPix(im, p, (win, win), (-1,-1), (EPS + MAX_ITER, subpix_iter, subpix_eps))                                   (:113-115)
 *   p = concatenate((q, p))                                                                                                    (:116)
 *   t, R, res = estimateWorldCameraPose(K, q, plate, findR=True)          (:118; plate_host: worldPointsLicensePlate, 4 x 3 float64, host)
 *   p3 = addcol0(image2world(K, R, t, p)) @ R + t                                                                              (:119)
 *   vp = insidebbox(p, boxa)                                                                                                   (:126, images.py:22-27)
 * Outputs (device, sized for 4 + max_corners tracks): p_out [.. x 2] float32, p3_out [.. x 3] float64, vp_out uint8, t_out float[3], R_out double[9]
 * (float32-rounded like NLS.py:180), res_out double[1], n_out int[1] = 4 + corners found (rows beyond it: vp 0, p3 0).  roi_host (may be NULL): host
 * int[8] = boxa, boxb.  The outputs feed vh_session_init directly. */
VH_API int vh_frame0_init(vh_ctx* ctx, const uint8_t* im, int w, int h, int stride, const float* q_host, const double* K_host,
                          const double* plate_host, int border_x, int border_y, int max_corners, double quality, int block, double k,
                          int subpix_win, int subpix_iter, double subpix_eps, float* p_out, double* p3_out, uint8_t* vp_out, float* t_out,
                          double* R_out, double* res_out, int* n_out, int* roi_host, void* stream);

/* ---- tracker session: the frame loop body of vidExample.py:133-160 on the device, for ctx->batch streams ------- */
/* device pointers into the state of one stream (read with vh_copy_to_host / torch) */
typedef struct {
    const uint8_t* vg;      /* N0      global validity mask (vidExample.py:125,135)            */
    const uint8_t* vp;      /* N0      pose-track mask (vidExample.py:126,136,160)             */
    const float* p;         /* n_cur x 2 compacted current points                              */
    const int* ids;         /* n_cur   global ids of the rows of p                             */
    const double* p3;       /* N0 x 3  world points                                            */
    const float* P;         /* history (vidExample.py:128-129,151-153), FRAME-MAJOR [nhist][5][N0]: entry (row, track, frame) at
                             * P[(frame * 5 + row) * N0 + track]; the reference's array is its transpose to [5,N0,nhist]  */
    const float* B;         /* [nhist,14]  (vidExample.py:44,142-146)                          */
    const float* S;         /* [nhist,9]   (vidExample.py:45,164)                              */
    const int* n_cur;
    const int* n_pose;
    const float* t;         /* 3       last pose translation                                   */
    const double* res;      /* 1       last rms reprojection residual                          */
    const int* frame_i;
    const int* klt_flags;
    const int* pose_info;   /* 2       iterations, converged                                   */
    const int* sel_pw;      /* n_pose  global ids of the pose tracks                           */
    const double* p_proj;   /* n_pose x 2                                                      */
    /* (vh_version >= 104) layout of P in floats: entry (row, track, frame) at P[row * P_row_stride + track * P_track_stride + frame * P_frame_stride].
     * Today (N0, 1, 5 N0) = frame-major [nhist][5][N0]; the reference's [5,N0,nhist] would read (N0 nhist, nhist, 1).  A C consumer indexes
     * through these, never through a remembered layout (version 103 changed it from the reference's, silently for such a consumer). */
    size_t P_row_stride, P_track_stride, P_frame_stride;
    int n0, nhist;
} vh_session_view;

/* K_host: 9 doubles; k_is_float32 != 0: the caller's K array is float32 (vidExample.py:29-33), so fcnMSV1_t builds its rays in
 * float32 like numpy does.  n0 = number of initial tracks, nhist = number of frames of history (P, B, S rows),
 * msv_frame = frame index at which fcnMSV1_t re-triangulates (vidExample.py:155; <= 0 disables; a frame the history reaches -- < nhist -- must be below
 * 2048: -1 otherwise, the limit of vh_msv1_t). */
VH_API int vh_session_create(vh_session** out, vh_ctx* ctx, int n0, int nhist, int w, int h, const double* K_host,
                             int k_is_float32, const vh_lk_params* coarse_host, const vh_lk_params* fine_host, int msv_frame);
VH_API void vh_session_destroy(vh_session* s);
/* frame-0 state of stream `slot` (vidExample.py:116-131): p n0 x 2, p3 n0 x 3, vp n0 (device); t0_host = plate pose t */
VH_API int vh_session_init(vh_session* s, int slot, const uint8_t* frame0, int stride, const float* p, const double* p3,
                           const uint8_t* vp, const float* t0_host, float time0, float frame_no, float res0, void* stream);
/* the same straight from the outputs of vh_frame0_init, all still on the device (no read-back between frame 0 and the loop): t0_dev float[3], res0_dev
 * double[1], n_dev int[1] = number of frame-0 tracks (may be NULL: n0).  With *n_dev < n0 the session runs the first *n_dev rows; the rest are tracks
 * that never existed (vg = 0, history NaN, S[0,2] = *n_dev). */
VH_API int vh_session_init_dev(vh_session* s, int slot, const uint8_t* frame0, int stride, const float* p, const double* p3, const uint8_t* vp,
                               const float* t0_dev, const double* res0_dev, const int* n_dev, float time0, float frame_no, void* stream);
/* one frame for every stream: frames_dev = device array of ctx->batch frame pointers (dense, w x h) */
VH_API int vh_session_step(vh_session* s, const uint8_t* const* frames_dev, float time_s, float frame_no, void* stream);
/* the same with one timestamp / frame number PER STREAM (device float[batch] each): independent videos with their own
 * CAP_PROP_POS_MSEC / frame counters (vidExample.py:142: B[i,12:14]).  The frames of step i must stay alive until step i+1 has
 * run (they are that step's im0). */
VH_API int vh_session_step_v(vh_session* s, const uint8_t* const* frames_dev, const float* time_s_dev, const float* frame_no_dev,
                             void* stream);
VH_API int vh_session_ptrs(vh_session* s, int slot, vh_session_view* out_host);
/* Packed track state of every stream for the cross-GPU exchange (RCCL all-gather, DESIGN.md "multi-GPU"):
 * out = device float32 [batch][8 + 3*n0]: {n_cur, n_pose, frame_i, klt_flags, t[3], res | p (n0 x 2) | ids (n0, int32 bits)} */
/* Fused frame ingest for every stream of the session (vidExample.py:91 + KLT.py:111-113 in ONE pass over the BGR frames): bgr_frames_dev / gray_frames_dev =
 * device arrays of ctx->batch pointers (h x w x 3 with rows bgr_stride bytes apart; dense w x h gray destinations owned by the caller, a null BGR pointer skips
 * the stream).  Writes the gray frames and, straight into the session, the quarter-scale images the next vh_session_step starts from (that step then
 * skips its own resize); pass the same gray frames to that step. */
VH_API int vh_session_ingest_bgr(vh_session* s, const uint8_t* const* bgr_frames_dev, int bgr_stride, uint8_t* const* gray_frames_dev, void* stream);
VH_API int vh_session_pack_state(vh_session* s, float* out, void* stream);

/* The vh_debug_* switches below are PROCESS-WIDE (atomics): they re-route every context of the process, and exist for the parity tests and A/B
 * experiments only -- every route they select is bit-identical to the default one.  Not a per-stream control; do not flip them in a multi-tenant process
 * for anything but a test.
 * test hook: route every LK window through one implementation -- 1: per-sample kernel, 2: strip kernel, 3: LDS-staged
 * kernel, 4: 4-tracks-per-wavefront kernel (15x15 windows; others as in 2), 5 / 6 / 7: LDS-staged 51x51 kernel with 1 / 2 / 4
 * wavefronts per track (other windows: default routing), 0: default routing per window and load.  All are bit-identical. */
VH_API void vh_debug_force_generic_lk(int on);

/* test hook (vh_version >= 105): launch slots one workgroup of the one-wavefront LDS-staged LK kernel (k_lk3<.., 1, ..>, routes 3 and 5) solves one after the
 * other -- 0: chosen by load (1 below 49 152 tracks in flight, 2 from there, 4 from 98 304), n > 0: n (clamped to 8).  Bit-identical results. */
VH_API void vh_debug_lk3_tpw(int n);

/* test hook: estimateAffine2D stand-in -- 1: always the three-kernel path (hypotheses spread over the chip), 2: the fused
 * one-workgroup-per-stream kernel whenever the problem fits it (latency path), 0: default routing by batch size.  Bit-identical results. */
VH_API void vh_debug_ransac_path(int mode);

/* test hook: 1 accumulates the reduced camera system of vh_nls_batch on the VALU instead of the f64 matrix cores */
VH_API void vh_debug_ba_force_valu(int on);
/* test hook: pyrDown with 2 / 4 / 8 output rows per thread whatever the launch size (0: chosen by size) */
VH_API void vh_debug_pyr_rows(int rows);
/* test hook: the LK launches of vh_klt_main / vh_session_step walk a stream's tracks in spatial order (a counting sort of the previous positions by
 * 64 x 32-px cell, k_klt_setup) -- 1: always, 0: never, -1: default (from 24000 tracks in flight).  Launch order only: outputs keep the caller's
 * indices and every result is bit-identical. */
VH_API void vh_debug_klt_order(int mode);

/* ---- measurement aids (bench.py): HIP-event timing of the LK launches + Newton-iteration statistics ------------- */
VH_API int vh_profile_begin(vh_ctx* ctx, int max_launches);
/* all outputs host arrays of 3 (KLTmain stage 0: quarter scale, 1: coarse ROI, 2: fine) */
VH_API int vh_profile_end(vh_ctx* ctx, double* ms_sum, int* launches, unsigned long long* iters, unsigned long long* setups);
/* every profiled stage of the same run: 0-2 the LK launches of KLTmain, 3 ROI warp, 4 pyrDown (+ border ring), 5 RANSAC, 6 quarter-scale resize,
 * 7 session frame kernel, 8-12 bundle adjustment (Jacobian, Schur complement, reduction, solve, update).  ms_sum / launches: nstages entries. */
VH_API int vh_profile_detail(vh_ctx* ctx, int all_stages);  /* before vh_profile_begin: 1 = every stage (default), 0 = the LK launches only */
VH_API int vh_profile_end_stages(vh_ctx* ctx, int nstages, double* ms_sum, int* launches);
/* the kernel each of the three LK launches of the last vh_klt_main / vh_session_step took (the launcher's own routing decision, so nobody mirrors its
 * thresholds): routes_host int[3] (ids of vh_debug_force_generic_lk), names_host char[3][32] (may be NULL) */
VH_API int vh_profile_lk_routes(vh_ctx* ctx, int* routes_host, char* names_host);
/* (vh_version >= 105) launch slots per workgroup of the same three launches (see vh_debug_lk3_tpw): tpw_host int[3]; a stateless vh_pyr_lk call reports
 * its route and slot count through entry 0 (like its iteration counters) */
VH_API int vh_profile_lk_tpw(vh_ctx* ctx, int* tpw_host);
/* host copy of every stream's ROI (x0, x1, y0, y1) of the last KLTmain call (images.py:9-19 as KLT.py:121-123 applies it): 4 ints per stream */
VH_API int vh_klt_rois(vh_ctx* ctx, int* roi_host);

#ifdef __cplusplus
}
#endif
#endif /* VELOCITY_HIP_H */
