// Bundle-adjustment job descriptors (internal).
#pragma once
#include "vh_common.hpp"

struct BaJob {  // passed by value to every BA kernel
    double K[9];
    const double* z;  // [2 * nt * (nc+1)] measurements, [all u | all v], camera-major / track-minor (NLS.py:198-199)
    double* x;        // model 0: [3 nt + 6 nc] points | camera positions | camera rpy (NLS.py:203)
                      // model 1: [3 nt + 3 + 2 + nc] points | joint rpy | el, az | camera ranges (NLS.py:274)
    double* camR;     // model 0: [(nc+1)][4][9] R(rpy) and its three forward-difference neighbours
                      // model 1: [4][9] joint R and neighbours, then [(nc+1)][4][3] camera offsets: base, +d el, +d az, +d range
    double* r;        // [m][2] residuals z - zhat
    double* Jp;       // [m][2][3] d(u,v)/d(point)
    double* Jc;       // [m][2][6] d(u,v)/d(camera pos, rpy)
    double* tp;       // [nt][3]  (U+I)^-1 gp
    double* Lc;       // [nt][6]  lower-triangular Cholesky factor of (U+I)^-1 = L L^T (l00 l10 l11 l20 l21 l22): matrix-core path only
    double* Y;        // [nt][6 nc][3]  (U+I)^-1 W   (matrix-core path: Z = L^T W instead, zmode = 1)
    double* Spart;    // [nparts][(6 nc)^2]
    double* Rpart;    // [nparts][6 nc]  (43+ cameras: [max(nparts, BA_ZB_MAX)][6 nc], one partial per k_ba_zbuild workgroup)
    double* Dpart;    // 43+ cameras only: [BA_ZB_MAX][36 nc] partial diagonal blocks V_c of the k_ba_zbuild workgroups
    double* Sfull;    // [6 nc][6 nc + 1]
    double* dc;       // [6 nc]
    double* acc;      // [2] sum r^2, sum delta^2
    double* rslot;    // [16] partial sums of r^2 (k_ba_jac spreads its atomics, k_ba_reduce folds them into acc[0])
    double* trace;    // [max_iter][2] rms(z - zhat), rms(delta) (what NLS.py:238 prints)
    int* info;        // [2] iterations, converged
    int* done;
    unsigned* ticket;
    double nx_total, nz_total;  // normalisation of rms(delta) / rms(residual): whole-problem sizes (sharded runs)
    int nt, nc;
    int nq;     // reduced unknowns: 6 nc (model 0) or nc + 5 (model 1)
    int model;  // 0: fcnNLS_batch (free cameras, NLS.py:186-250); 1: fcnNLS_batch2 (joint rotation + straight-line trajectory, NLS.py:253-328)
    int add_identity, count_cams, defer_finalize;
    int dbg;    // experiment switches (VH_BA_DBG): 1 skip MFMA, 2 skip Z / diag VALU, 4 skip global fetch, 8 skip barrier, 64 VALU Gauss-Jordan solve, 128 elimination kernels above 128 unknowns, 256 right-looking Cholesky (two launches per panel)
    int zmode;  // 1: Y holds Z = L^T W and Spart holds only the upper-triangle 16x16 tiles (k_ba_schur_mfma); 0: Y = (U+I)^-1 W, full Spart
    // batched independent windows (vh_nls_batch_multi): blockIdx.y selects the window; every pointer above is window 0's
    int nwin;
    size_t ws_stride;                // bytes between the workspaces of consecutive windows
    size_t z_stride, x_stride;       // doubles between consecutive windows' z / x
    size_t trace_stride;             // doubles
    size_t info_stride;              // ints
};

struct BaProblem {
    double K[9];
    const double* z;
    double* x;
    double* trace;
    int* info;
    void* workspace;
    int nt, nc, max_iter, nparts;
    int phase, it;   // phase -1: whole solve; 0..3: the pieces of one sharded iteration (see vh_ba_run)
    int add_identity, count_cams, defer_finalize;
    double nx_total, nz_total;
    int model;       // see BaJob::model
    int force_valu;  // test hook: 1 = accumulate the reduced camera system on the VALU instead of the matrix cores
    int nwin;        // independent windows solved by the same launches (>= 1); z, x, trace, info and workspace are arrays of nwin
    size_t ws_stride, z_stride, x_stride, trace_stride, info_stride;  // see BaJob (ignored when nwin == 1)
    struct vh_ctx* ctx;  // may be null: per-kernel HIP-event timing when its profiling is on (vh_profile_begin)
    void** graph_cache;  // may be null: where the owner (vh_ctx) keeps the replayable launch sequences of whole solves (vh_ba_graph_cache_free)
};
void vh_ba_graph_cache_free(void* cache);

size_t vh_ba_workspace_bytes(int nt, int nc, int nparts);
int vh_ba_run(const BaProblem& P, hipStream_t s);
void vh_ba_exchange_span(const BaProblem& P, size_t* offset_bytes, size_t* n_doubles);
