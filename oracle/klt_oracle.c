/*
 * CPU oracle for the KLT half of the hot path.  TEST INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * Restates, in plain C, the algorithm behind the reference's tracker:
 *     utils/KLT.py:37-51   cv2calcOpticalFlowPyrLK  (LK + forward-backward gate)
 *     utils/KLT.py:55-95   KLTregional              (ROI crop / shift / affine remap, LK, map back)
 *     utils/KLT.py:99-134  KLTmain                  (1/4-scale LK -> RANSAC -> ROI LK -> RANSAC affine -> fine LK)
 *     utils/images.py:9-19 boundingRect
 * All arithmetic of those functions lives in an un-pinned third-party dependency (requirements.txt:5,
 * `opencv-python`, no version, absent from this image and from /root/reference).  What is restated here
 * is OpenCV 4.x's PUBLISHED algorithm: pyrDown [1 4 6 4 1]^2/256 with REFLECT_101,
 * un-normalised int16 Scharr derivatives, 14-bit fixed-point bilinear windows, the min-eigenvalue test,
 * the Newton iteration with its two stop rules, remap's 5-bit fixed-point bilinear, nearest resize.
 *
 * PARITY UNPINNED: the reference holds no KLT test, golden vector or fixture and cv2 cannot run here, so this
 * restatement is pinned only by analytic-flow ground truth (tests/test_oracle_klt.py) and by self-consistency.
 * Deliberate, documented choices where OpenCV is build/SIMD dependent or not reproducible:
 *   - window sums (A11,A12,A22,b1,b2,err) are accumulated EXACTLY in int64 and converted once to float
 *     (OpenCV: float or int32-lane accumulation depending on the SIMD path) -> order independent;
 *   - estimateAffine2D's RANSAC uses our own counter-based RNG (OpenCV's sequence is not reproducible
 *     without its generator) and the LM refinement is replaced by the linear least-squares fit it
 *     converges to; reductions use int64 fixed point so they are order independent;
 *   - the two crashing lines of the reference (KLT.py:87, vidExample.py:134) follow SURVEY Appendix B intent.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define KO_API __attribute__((visibility("default")))

/* worker threads of every parallel loop below (bench.py's cpu_baseline: all host cores / one core).  n <= 0: all cores. */
KO_API int ko_set_threads(int n)
{
#ifdef _OPENMP
    if (n <= 0) n = omp_get_num_procs();
    omp_set_num_threads(n);
    return n;
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------ */
/* sensitivity modes (tests/test_oracle_klt_sensitivity.py): the parity contract is mode 0 everywhere */
/* ------------------------------------------------------------------------------------------------ */
/* LK window sums (A11, A12, A22, b1, b2):
 *   0  exact int64 sums converted once to float32 -- THE contract (order independent, what the HIP kernels reproduce bit for bit);
 *   1  OpenCV's plain C path of lkpyramid.cpp (`typedef float acctype, itemtype`): float32 accumulators, raster order,
 *      `iA11 += (itemtype)(ixval * ixval)`, `ib1 += (itemtype)(diff * dIptr[0])`;
 *   2  the accumulation STRUCTURE of OpenCV's 128-bit universal-intrinsics path as recalled from upstream (not verifiable here):
 *      A: 4 float lanes (pixel x mod 4) of float(ix) * float(iy) multiply-adds over groups of 4 pixels, scalar float tail, lanes
 *      summed at the end; b: groups of 8 pixels, exact int32 sums of the pixel pairs (x, x+4) converted to float and added to
 *      float lanes, scalar float tail.
 * Modes 1 and 2 exist only to BOUND how much the exact-integer choice can move results relative to a float-accumulating OpenCV. */
static int g_accum_mode = 0;
KO_API void ko_set_accum_mode(int m) { g_accum_mode = m; }
/* estimateAffine2D refinement: 0 = closed-form least squares on the inliers (contract); 1 = OpenCV's sequence: the best 3-point
 * model refined by 10 iterations of its LMSolver (calib3d levmarq.cpp: lambda, R ratio test) on the inliers */
static int g_refine_mode = 0;
KO_API void ko_set_refine_mode(int m) { g_refine_mode = m; }

/* ------------------------------------------------------------------------------------------------ */
/* small helpers                                                                                     */
/* ------------------------------------------------------------------------------------------------ */
static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
static inline int ifloor(float v) { return (int)floorf(v); }
static inline int iround(float v) { return (int)lrintf(v); } /* round-half-even, like cvRound */
static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

typedef struct {
    int w, h;      /* logical size */
    int border;    /* padding on every side */
    int stride;    /* elements per row of the padded buffer */
    uint8_t *img;  /* padded, REFLECT_101 filled; pixel (x,y) at img[(y+border)*stride + x+border] */
    int16_t *der;  /* padded interleaved (Ix,Iy), constant 0 outside; may be NULL */
} level_t;

static void level_free(level_t *L)
{
    free(L->img);
    free(L->der);
    L->img = NULL;
    L->der = NULL;
}

/* copy a w x h image into a padded REFLECT_101 buffer */
static void level_from_image(level_t *L, const uint8_t *src, int w, int h, ptrdiff_t sstride, int border)
{
    L->w = w; L->h = h; L->border = border; L->stride = w + 2 * border;
    L->img = (uint8_t *)malloc((size_t)L->stride * (h + 2 * border));
    L->der = NULL;
#pragma omp parallel for schedule(static)
    for (int y = -border; y < h + border; y++) {
        const uint8_t *srow = src + (ptrdiff_t)reflect101(y, h) * sstride;
        uint8_t *drow = L->img + (size_t)(y + border) * L->stride;
        for (int x = -border; x < w + border; x++) drow[x + border] = srow[reflect101(x, w)];
    }
}

/* pyrDown: dst (w+1)/2 x (h+1)/2, 5x5 [1 4 6 4 1]^2, REFLECT_101, (sum+128)>>8 */
static void pyr_down_raw(const uint8_t *src, int w, int h, ptrdiff_t sstride, uint8_t *dst, int dw, int dh, ptrdiff_t dstride)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++) {
        const uint8_t *r[5];
        for (int k = 0; k < 5; k++) r[k] = src + (ptrdiff_t)reflect101(2 * y - 2 + k, h) * sstride;
        for (int x = 0; x < dw; x++) {
            int c[5];
            for (int k = 0; k < 5; k++) c[k] = reflect101(2 * x - 2 + k, w);
            int acc = 0;
            static const int wv[5] = {1, 4, 6, 4, 1};
            for (int k = 0; k < 5; k++) {
                const uint8_t *row = r[k];
                int hsum = row[c[0]] + 4 * row[c[1]] + 6 * row[c[2]] + 4 * row[c[3]] + row[c[4]];
                acc += wv[k] * hsum;
            }
            dst[(ptrdiff_t)y * dstride + x] = (uint8_t)((acc + 128) >> 8);
        }
    }
}

KO_API void ko_pyr_down(const uint8_t *src, int w, int h, int sstride, uint8_t *dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    pyr_down_raw(src, w, h, sstride, dst, dw, dh, dw);
}

/* resize(fx=fy=0.25, INTER_NEAREST): dsize = round-half-even(src*0.25), sx = min(4x, W-1)  (KLT.py:111-113) */
KO_API void ko_resize_quarter_dims(int w, int h, int *dw, int *dh)
{
    *dw = (int)lrint(w * 0.25);
    *dh = (int)lrint(h * 0.25);
}
KO_API void ko_resize_quarter(const uint8_t *src, int w, int h, int sstride, uint8_t *dst)
{
    int dw, dh;
    ko_resize_quarter_dims(w, h, &dw, &dh);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++) {
        int sy = 4 * y < h - 1 ? 4 * y : h - 1;
        for (int x = 0; x < dw; x++) {
            int sx = 4 * x < w - 1 ? 4 * x : w - 1;
            dst[(size_t)y * dw + x] = src[(ptrdiff_t)sy * sstride + sx];
        }
    }
}

/* cv2.resize(im, (0,0), fx, fy, INTER_NEAREST) (vidExample.py:99-102, the `scale != 1` ingest branch): dsize = round(w fx) x round(h fy),
 * sx = min(floor(x / fx), w - 1) (OpenCV resizeNN: x_ofs = cvFloor(x * (1/fx))) */
KO_API void ko_resize_nearest_dims(int w, int h, double fx, double fy, int *dw, int *dh)
{
    *dw = (int)lrint(w * fx);
    *dh = (int)lrint(h * fy);
}
KO_API void ko_resize_nearest(const uint8_t *src, int w, int h, int sstride, double fx, double fy, uint8_t *dst)
{
    int dw, dh;
    ko_resize_nearest_dims(w, h, fx, fy, &dw, &dh);
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    for (int y = 0; y < dh; y++) {
        int sy = (int)floor(y * ify);
        if (sy > h - 1) sy = h - 1;
        for (int x = 0; x < dw; x++) {
            int sx = (int)floor(x * ifx);
            if (sx > w - 1) sx = w - 1;
            dst[(size_t)y * dw + x] = src[(ptrdiff_t)sy * sstride + sx];
        }
    }
}

/* Scharr derivative of the un-padded image with REFLECT_101 support, zero in the padding */
static void level_make_deriv(level_t *L)
{
    int b = L->border, w = L->w, h = L->h, st = L->stride;
    L->der = (int16_t *)calloc((size_t)st * (h + 2 * b) * 2, sizeof(int16_t));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        /* rows/cols outside the image come from the REFLECT_101 padding (border >= 1) */
        const uint8_t *r0 = L->img + (size_t)(y - 1 + b) * st + b;
        const uint8_t *r1 = r0 + st, *r2 = r1 + st;
        int16_t *d = L->der + ((size_t)(y + b) * st + b) * 2;
        for (int x = 0; x < w; x++) {
            /* vertical smooth [3 10 3] and vertical diff [-1 0 1] at columns x-1, x, x+1 */
            int s_m = (r0[x - 1] + r2[x - 1]) * 3 + r1[x - 1] * 10;
            int s_p = (r0[x + 1] + r2[x + 1]) * 3 + r1[x + 1] * 10;
            int d_m = r2[x - 1] - r0[x - 1], d_c = r2[x] - r0[x], d_p = r2[x + 1] - r0[x + 1];
            d[2 * x] = (int16_t)(s_p - s_m);
            d[2 * x + 1] = (int16_t)((d_p + d_m) * 3 + d_c * 10);
        }
    }
}

/* pyramid with OpenCV's level truncation: stop when the NEXT level would be <= win in either dim */
typedef struct { int nlevels; level_t lv[16]; } pyramid_t;

static void pyramid_build(pyramid_t *P, const uint8_t *src, int w, int h, ptrdiff_t sstride, int win, int max_level, int with_deriv)
{
    P->nlevels = 0;
    level_from_image(&P->lv[0], src, w, h, sstride, win);
    int lw = w, lh = h;
    for (int level = 0; level <= max_level; level++) {
        if (level > 0) {
            level_t *prev = &P->lv[level - 1];
            uint8_t *tmp = (uint8_t *)malloc((size_t)lw * lh);
            pyr_down_raw(prev->img + (size_t)prev->border * prev->stride + prev->border, prev->w, prev->h, prev->stride, tmp, lw, lh, lw);
            level_from_image(&P->lv[level], tmp, lw, lh, lw, win);
            free(tmp);
        }
        if (with_deriv) level_make_deriv(&P->lv[level]);
        P->nlevels = level + 1;
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;
    }
}
static void pyramid_free(pyramid_t *P)
{
    for (int i = 0; i < P->nlevels; i++) level_free(&P->lv[i]);
    P->nlevels = 0;
}

KO_API int ko_pyramid_levels(int w, int h, int win, int max_level)
{
    int n = 0;
    for (int level = 0; level <= max_level; level++) {
        n = level + 1;
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) break;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* pyramidal Lucas-Kanade                                              */
/* ------------------------------------------------------------------------------------------------ */
#define W_BITS 14
static const float FLT_SCALE = 1.f / (1 << 20);

static inline void bilinear_weights(float a, float b, int *w00, int *w01, int *w10, int *w11)
{
    *w00 = iround((1.f - a) * (1.f - b) * (1 << W_BITS));
    *w01 = iround(a * (1.f - b) * (1 << W_BITS));
    *w10 = iround((1.f - a) * b * (1 << W_BITS));
    *w11 = (1 << W_BITS) - *w00 - *w01 - *w10;
}

/* diagnostic hook (tools/exp/lk_bucketing_study.py): Newton iterations of every point on every level of the NEXT lk runs, buf[(run * n + i) * 8 + level] */
static int *g_iter_trace = 0;
static int g_iter_run = 0;
static __thread int t_last_iters;
KO_API void ko_set_iter_trace(int *buf) { g_iter_trace = buf; g_iter_run = 0; }

/* one point on one level; returns nothing, updates next[2], status, err */
static void lk_point_level(const level_t *I, const level_t *J, int win, int level, int top_level, int max_count, double eps2,
                           float min_eig_thr, const float prev_pt_full[2], float next[2], uint8_t *status, float *err,
                           int16_t *Iwin, int16_t *dIwin)
{
    const float half = (win - 1) * 0.5f;
    const float lscale = (float)(1. / (1 << level));
    float px = prev_pt_full[0] * lscale, py = prev_pt_full[1] * lscale;
    float nx, ny;
    t_last_iters = 0;
    if (level == top_level) { nx = px; ny = py; }
    else { nx = next[0] * 2.f; ny = next[1] * 2.f; }
    next[0] = nx; next[1] = ny;

    px -= half; py -= half;
    int ipx = ifloor(px), ipy = ifloor(py);
    if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
        if (level == 0) { *status = 0; *err = 0.f; }
        return;
    }
    int w00, w01, w10, w11;
    bilinear_weights(px - ipx, py - ipy, &w00, &w01, &w10, &w11);

    const int sI = I->stride, sD = I->stride * 2, sJ = J->stride;
    int64_t sA11 = 0, sA12 = 0, sA22 = 0;
    const int amode = g_accum_mode;
    float fA11 = 0.f, fA12 = 0.f, fA22 = 0.f;                 /* modes 1, 2: scalar float accumulators (mode 2: the row tails) */
    float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0}; /* mode 2: lanes */
    for (int y = 0; y < win; y++) {
        const uint8_t *src = I->img + (size_t)(y + ipy + I->border) * sI + ipx + I->border;
        const int16_t *ds = I->der + ((size_t)(y + ipy + I->border) * sI + ipx + I->border) * 2;
        for (int x = 0; x < win; x++) {
            int iv = descale(src[x] * w00 + src[x + 1] * w01 + src[x + sI] * w10 + src[x + sI + 1] * w11, W_BITS - 5);
            int ix = descale(ds[2 * x] * w00 + ds[2 * x + 2] * w01 + ds[2 * x + sD] * w10 + ds[2 * x + sD + 2] * w11, W_BITS);
            int iy = descale(ds[2 * x + 1] * w00 + ds[2 * x + 3] * w01 + ds[2 * x + sD + 1] * w10 + ds[2 * x + sD + 3] * w11, W_BITS);
            Iwin[y * win + x] = (int16_t)iv;
            dIwin[2 * (y * win + x)] = (int16_t)ix;
            dIwin[2 * (y * win + x) + 1] = (int16_t)iy;
            sA11 += (int64_t)ix * ix;
            sA12 += (int64_t)ix * iy;
            sA22 += (int64_t)iy * iy;
            if (amode == 1 || (amode == 2 && x >= (win & ~3))) {
                fA11 += (float)(ix * ix); fA12 += (float)(ix * iy); fA22 += (float)(iy * iy);
            } else if (amode == 2) {
                const float fx = (float)ix, fy = (float)iy;
                qA11[x & 3] = fx * fx + qA11[x & 3]; qA12[x & 3] = fx * fy + qA12[x & 3]; qA22[x & 3] = fy * fy + qA22[x & 3];
            }
        }
    }
    float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
    if (amode == 2) {
        fA11 += qA11[0] + qA11[1] + qA11[2] + qA11[3]; fA12 += qA12[0] + qA12[1] + qA12[2] + qA12[3]; fA22 += qA22[0] + qA22[1] + qA22[2] + qA22[3];
    }
    if (amode) { A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE; }
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if (minEig < min_eig_thr || D < 1.1920929e-07f) {
        if (level == 0) *status = 0;
        return;
    }
    D = 1.f / D;

    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    for (int j = 0; j < max_count; j++) {
        int inx = ifloor(nx), iny = ifloor(ny);
        if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
            if (level == 0) *status = 0;
            break;
        }
        bilinear_weights(nx - inx, ny - iny, &w00, &w01, &w10, &w11);
        t_last_iters++;
        int64_t sb1 = 0, sb2 = 0;
        float fb1 = 0.f, fb2 = 0.f, qb1[4] = {0, 0, 0, 0}, qb2[4] = {0, 0, 0, 0};
        int dbuf[8];
        for (int y = 0; y < win; y++) {
            const uint8_t *jp = J->img + (size_t)(y + iny + J->border) * sJ + inx + J->border;
            for (int x = 0; x < win; x++) {
                int diff = descale(jp[x] * w00 + jp[x + 1] * w01 + jp[x + sJ] * w10 + jp[x + sJ + 1] * w11, W_BITS - 5) - Iwin[y * win + x];
                sb1 += (int64_t)diff * dIwin[2 * (y * win + x)];
                sb2 += (int64_t)diff * dIwin[2 * (y * win + x) + 1];
                if (amode == 1 || (amode == 2 && x >= (win & ~7))) {
                    fb1 += (float)(diff * dIwin[2 * (y * win + x)]);
                    fb2 += (float)(diff * dIwin[2 * (y * win + x) + 1]);
                } else if (amode == 2) {
                    dbuf[x & 7] = diff;
                    if ((x & 7) == 7) { /* a full group of 8: pixel pairs (k, k+4) summed exactly in int32, then float lanes */
                        const int16_t *d = dIwin + 2 * (y * win + x - 7);
                        for (int k = 0; k < 4; k++) {
                            qb1[k] += (float)(dbuf[k] * d[2 * k] + dbuf[k + 4] * d[2 * (k + 4)]);
                            qb2[k] += (float)(dbuf[k] * d[2 * k + 1] + dbuf[k + 4] * d[2 * (k + 4) + 1]);
                        }
                    }
                }
            }
        }
        float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
        if (amode == 2) { fb1 += (qb1[0] + qb1[2]) + (qb1[1] + qb1[3]); fb2 += (qb2[0] + qb2[2]) + (qb2[1] + qb2[3]); }
        if (amode) { b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE; }
        float dx = (A12 * b2 - A22 * b1) * D;
        float dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        next[0] = nx + half; next[1] = ny + half;
        if ((double)dx * dx + (double)dy * dy <= eps2) break;
        if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
            next[0] -= dx * 0.5f; next[1] -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }

    if (*status && level == 0) {
        float fx = next[0] - half, fy = next[1] - half;
        int inx = ifloor(fx), iny = ifloor(fy);
        if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) { *status = 0; return; }
        bilinear_weights(fx - inx, fy - iny, &w00, &w01, &w10, &w11);
        int64_t se = 0;
        for (int y = 0; y < win; y++) {
            const uint8_t *jp = J->img + (size_t)(y + iny + J->border) * sJ + inx + J->border;
            for (int x = 0; x < win; x++) {
                int diff = descale(jp[x] * w00 + jp[x + 1] * w01 + jp[x + sJ] * w10 + jp[x + sJ + 1] * w11, W_BITS - 5) - Iwin[y * win + x];
                se += diff < 0 ? -diff : diff;
            }
        }
        *err = (float)se * (1.f / (float)(32 * win * win));
    }
}

static void criteria_clamp(int *max_count, double *eps)
{
    if (*max_count < 0) *max_count = 0;
    if (*max_count > 100) *max_count = 100;
    if (*eps < 0) *eps = 0;
    if (*eps > 10) *eps = 10;
    *eps = *eps * *eps;
}

static void lk_run(const pyramid_t *PI, const pyramid_t *PJ, const float *prev_pts, int n, int win, int max_count, double eps,
                   float *next_pts, uint8_t *status, float *err)
{
    criteria_clamp(&max_count, &eps);
    int nl = PI->nlevels < PJ->nlevels ? PI->nlevels : PJ->nlevels;
#pragma omp parallel
    {
        int16_t *Iwin = (int16_t *)malloc(sizeof(int16_t) * win * win);
        int16_t *dIwin = (int16_t *)malloc(sizeof(int16_t) * win * win * 2);
#pragma omp for schedule(dynamic, 16)
        for (int i = 0; i < n; i++) {
            status[i] = 1;
            err[i] = 0.f;
            float nxt[2] = {0.f, 0.f};
            for (int level = nl - 1; level >= 0; level--) {
                lk_point_level(&PI->lv[level], &PJ->lv[level], win, level, nl - 1, max_count, eps, 1e-4f, prev_pts + 2 * i, nxt,
                               &status[i], &err[i], Iwin, dIwin);
                if (g_iter_trace && level < 8) g_iter_trace[((size_t)g_iter_run * n + i) * 8 + level] = t_last_iters;
            }
            next_pts[2 * i] = nxt[0];
            next_pts[2 * i + 1] = nxt[1];
        }
        free(Iwin);
        free(dIwin);
    }
    if (g_iter_trace) g_iter_run++;
}

/* cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize=(win,win), maxLevel, criteria=(EPS|COUNT, max_count, eps)) */
KO_API void ko_pyr_lk(const uint8_t *prev, const uint8_t *next, int w, int h, int pstride, int nstride, const float *prev_pts, int n,
                      int win, int max_level, int max_count, double eps, float *next_pts, uint8_t *status, float *err)
{
    pyramid_t PI, PJ;
    pyramid_build(&PI, prev, w, h, pstride, win, max_level, 1);
    pyramid_build(&PJ, next, w, h, nstride, win, max_level, 0);
    lk_run(&PI, &PJ, prev_pts, n, win, max_count, eps, next_pts, status, err);
    pyramid_free(&PI);
    pyramid_free(&PJ);
}

/* cv2calcOpticalFlowPyrLK (KLT.py:37-51): forward LK; if fbt >= 0 backward LK from the results and
 * v = v & v2 & (||p1 - p1'|| < fbt), the norm and compare in float32 (common.py:13-15). */
KO_API void ko_lk_fb(const uint8_t *im1, const uint8_t *im2, int w, int h, int stride1, int stride2, const float *p1, int n, int win,
                     int max_level, int max_count, double eps, float fbt, float *p2, uint8_t *v, float *err, float *fbe_out)
{
    pyramid_t P1, P2;
    int fb = fbt >= 0.f;
    pyramid_build(&P1, im1, w, h, stride1, win, max_level, 1);
    pyramid_build(&P2, im2, w, h, stride2, win, max_level, fb);
    lk_run(&P1, &P2, p1, n, win, max_count, eps, p2, v, err);
    if (fb) {
        float *p1b = (float *)malloc(sizeof(float) * 2 * n);
        uint8_t *v2 = (uint8_t *)malloc(n);
        float *e2 = (float *)malloc(sizeof(float) * n);
        lk_run(&P2, &P1, p2, n, win, max_count, eps, p1b, v2, e2);
        for (int i = 0; i < n; i++) {
            float dx = p1[2 * i] - p1b[2 * i], dy = p1[2 * i + 1] - p1b[2 * i + 1];
            float fbe = sqrtf(dx * dx + dy * dy);
            if (fbe_out) fbe_out[i] = fbe;
            v[i] = (uint8_t)(v[i] && v2[i] && (fbe < fbt));
        }
        free(p1b); free(v2); free(e2);
    }
    pyramid_free(&P1);
    pyramid_free(&P2);
}

/* ------------------------------------------------------------------------------------------------ */
/* boundingRect (images.py:9-19 over cv2.boundingRect of float points)                               */
/* ------------------------------------------------------------------------------------------------ */
KO_API void ko_bounding_rect(const float *p, int n, int imw, int imh, int bx, int by, int roi[4])
{
    if (n <= 0) { /* no tracks left (the reference would raise on min() of an empty array; SURVEY App. B: resolve by intent): an empty ROI at the frame origin, as the product's glue */
        roi[0] = 1; roi[1] = 1; roi[2] = 1; roi[3] = 1;
        return;
    }
    float mnx = p[0], mxx = p[0], mny = p[1], mxy = p[1];
    for (int i = 1; i < n; i++) {
        float x = p[2 * i], y = p[2 * i + 1];
        if (x < mnx) mnx = x;
        if (x > mxx) mxx = x;
        if (y < mny) mny = y;
        if (y > mxy) mxy = y;
    }
    int x0 = ifloor(mnx), y0 = ifloor(mny);
    int bw = ifloor(mxx) - x0 + 1, bh = ifloor(mxy) - y0 + 1;
    int x1 = x0 + bw + bx, y1 = y0 + bh + by;
    x0 -= bx; y0 -= by;
    if (x0 < 1) x0 = 1;
    if (y0 < 1) y0 = 1;
    if (x1 > imw) x1 = imw;
    if (y1 > imh) y1 = imh;
    roi[0] = x0; roi[1] = x1; roi[2] = y0; roi[3] = y1;
}

/* ------------------------------------------------------------------------------------------------ */
/* affine remap of a ROI (KLT.py:70-73): map coords in float32, remap INTER_LINEAR, constant-0 border */
/* ------------------------------------------------------------------------------------------------ */
KO_API void ko_remap_affine(const uint8_t *im, int w, int h, int stride, const float T[6] /* 3x2 row-major */, int x0, int x1, int y0,
                            int y1, uint8_t *dst /* (y1-y0) x (x1-x0) */)
{
    int rw = x1 - x0;
#pragma omp parallel for schedule(static)
    for (int yy = y0; yy < y1; yy++) {
        float y = (float)yy;
        for (int xx = x0; xx < x1; xx++) {
            float x = (float)xx;
            float mx = (x * T[0] + y * T[2]) + T[4];
            float my = (x * T[1] + y * T[3]) + T[5];
            int fx = iround(mx * 32.f), fy = iround(my * 32.f);
            int sx = fx >> 5, sy = fy >> 5, ax = fx & 31, ay = fy & 31;
            int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
            int s00 = 0, s01 = 0, s10 = 0, s11 = 0;
            int inx0 = sx >= 0 && sx < w, inx1 = sx + 1 >= 0 && sx + 1 < w;
            int iny0 = sy >= 0 && sy < h, iny1 = sy + 1 >= 0 && sy + 1 < h;
            if (iny0 && inx0) s00 = im[(ptrdiff_t)sy * stride + sx];
            if (iny0 && inx1) s01 = im[(ptrdiff_t)sy * stride + sx + 1];
            if (iny1 && inx0) s10 = im[(ptrdiff_t)(sy + 1) * stride + sx];
            if (iny1 && inx1) s11 = im[(ptrdiff_t)(sy + 1) * stride + sx + 1];
            dst[(size_t)(yy - y0) * rw + (xx - x0)] = (uint8_t)((s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11 + (1 << 14)) >> 15);
        }
    }
}

/* shifted crop with zero padding outside the frame (KLT.py:65-68, Appendix B intent) */
KO_API void ko_crop_shift(const uint8_t *im, int w, int h, int stride, int x0, int x1, int y0, int y1, int dx, int dy, uint8_t *dst)
{
    int rw = x1 - x0;
    for (int yy = y0; yy < y1; yy++)
        for (int xx = x0; xx < x1; xx++) {
            int sx = xx + dx, sy = yy + dy;
            dst[(size_t)(yy - y0) * rw + (xx - x0)] = (sx >= 0 && sx < w && sy >= 0 && sy < h) ? im[(ptrdiff_t)sy * stride + sx] : 0;
        }
}

/* ------------------------------------------------------------------------------------------------ */
/* deterministic RANSAC affine (stands in for cv2.estimateAffine2D, KLT.py:116,127)                  */
/* ------------------------------------------------------------------------------------------------ */
#define RANSAC_MAX_ITERS 2000
#define RANSAC_THRESH 3.0
#define RANSAC_CONF 0.99
#define RANSAC_SEED 0x2545F491u

static inline uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
static inline uint32_t ransac_draw(uint32_t hyp, uint32_t k, uint32_t attempt, uint32_t m)
{
    uint32_t h = mix32(RANSAC_SEED ^ mix32(hyp * 0x9E3779B9u + k * 0x7F4A7C15u + attempt * 0x94D049BBu + 1u));
    return (uint32_t)(((uint64_t)h * m) >> 32);
}

/* log(x) for x > 0 from plain arithmetic only, so CPU and GPU agree bit for bit */
static double det_log(double x)
{
    union { double d; uint64_t u; } c;
    c.d = x;
    int e = (int)((c.u >> 52) & 0x7FF) - 1023;
    c.u = (c.u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull; /* m in [1,2) */
    double m = c.d;
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0), s2 = s * s, acc = 0.0;
    for (int k = 12; k >= 0; k--) acc = acc * s2 + 1.0 / (double)(2 * k + 1);
    return (double)e * 0.6931471805599453 + 2.0 * s * acc;
}

static int ransac_update_iters(double conf, double ep, int max_iters)
{
    double num = 1.0 - conf;
    if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
    double wgt = 1.0 - ep;
    double denom = 1.0 - wgt * wgt * wgt;
    if (denom < 2.2250738585072014e-308) return 0;
    num = det_log(num);
    denom = det_log(denom);
    if (denom >= 0 || -num >= max_iters * (-denom)) return max_iters;
    return (int)lrint(num / denom);
}

static int collinear(const double a[2], const double b[2], const double c[2])
{
    double dx1 = b[0] - a[0], dy1 = b[1] - a[1], dx2 = c[0] - a[0], dy2 = c[1] - a[1];
    return fabs(dx1 * dy2 - dy1 * dx2) <= 1.1920929e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
}

/* affine from three pairs; returns 0 when the sample is degenerate */
static int affine_from3(const float *from, const float *to, const int id[3], double M[6])
{
    double f[3][2], t[3][2];
    for (int k = 0; k < 3; k++) {
        f[k][0] = from[2 * id[k]]; f[k][1] = from[2 * id[k] + 1];
        t[k][0] = to[2 * id[k]];   t[k][1] = to[2 * id[k] + 1];
    }
    if (collinear(f[0], f[1], f[2]) || collinear(t[0], t[1], t[2])) return 0;
    double ax = f[0][0] - f[2][0], ay = f[0][1] - f[2][1], bx = f[1][0] - f[2][0], by = f[1][1] - f[2][1];
    double det = ax * by - bx * ay;
    if (det == 0.0) return 0;
    for (int r = 0; r < 2; r++) {
        double u0 = t[0][r] - t[2][r], u1 = t[1][r] - t[2][r];
        double a = (u0 * by - u1 * ay) / det;
        double b = (ax * u1 - bx * u0) / det;
        M[3 * r] = a;
        M[3 * r + 1] = b;
        M[3 * r + 2] = (t[2][r] - a * f[2][0]) - b * f[2][1];
    }
    return 1;
}

static inline int is_inlier(const double M[6], float x, float y, float u, float v)
{
    double ex = ((M[0] * x + M[1] * y) + M[2]) - u;
    double ey = ((M[3] * x + M[4] * y) + M[5]) - v;
    float e = (float)(ex * ex + ey * ey);
    return e <= (float)(RANSAC_THRESH * RANSAC_THRESH);
}

static int hypothesis(const float *from, const float *to, int m, uint32_t hyp, double M[6])
{
    int id[3];
    for (int k = 0; k < 3; k++) {
        int ok = 0;
        for (uint32_t a = 0; a < 16 && !ok; a++) {
            id[k] = (int)ransac_draw(hyp, (uint32_t)k, a, (uint32_t)m);
            ok = 1;
            for (int q = 0; q < k; q++) ok &= id[q] != id[k];
        }
        if (!ok) return 0;
    }
    return affine_from3(from, to, id, M);
}

static int64_t fixq(double v, int bits) { return (int64_t)llrint(ldexp(v, bits)); }

/* ---- OpenCV-style refinement of the RANSAC model (sensitivity mode, g_refine_mode == 1) ---------------------------------------- */
/* Affine2DRefineCallback: residuals (M x - u, per inlier, 2 rows) and their Jacobian w.r.t. the 6 entries of M (linear). */
static double lm_affine_residual(const float *from, const float *to, const uint8_t *inl, int m, const double h[6], double JtJ[36], double Jtr[6])
{
    double S = 0.0;
    if (JtJ) { memset(JtJ, 0, sizeof(double) * 36); memset(Jtr, 0, sizeof(double) * 6); }
    for (int i = 0; i < m; i++) {
        if (!inl[i]) continue;
        const double x = from[2 * i], y = from[2 * i + 1];
        const double ex = h[0] * x + h[1] * y + h[2] - to[2 * i], ey = h[3] * x + h[4] * y + h[5] - to[2 * i + 1];
        S += ex * ex + ey * ey;
        if (JtJ) {
            const double j[3] = {x, y, 1.0};
            for (int a = 0; a < 3; a++) {
                for (int b = 0; b < 3; b++) { JtJ[a * 6 + b] += j[a] * j[b]; JtJ[(a + 3) * 6 + b + 3] += j[a] * j[b]; }
                Jtr[a] += j[a] * ex; Jtr[a + 3] += j[a] * ey;
            }
        }
    }
    return S;
}
static int solve6(const double A_[36], const double b_[6], double x[6])
{
    double A[6][7];
    for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) A[r][c] = A_[r * 6 + c]; A[r][6] = b_[r]; }
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (A[p][c] == 0.0) return 0;
        if (p != c) for (int k = 0; k < 7; k++) { double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
        for (int r = 0; r < 6; r++) {
            if (r == c) continue;
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 7; k++) A[r][k] -= f * A[c][k];
        }
    }
    for (int r = 0; r < 6; r++) x[r] = A[r][6] / A[r][r];
    return 1;
}
/* OpenCV LMSolverImpl::run (calib3d/src/levmarq.cpp, 4.x): lambda starts at 1, diag scaling, gain-ratio test with Rlo/Rhi. */
static void lm_refine_affine(const float *from, const float *to, const uint8_t *inl, int m, double h[6], int max_iters)
{
    double A[36], v[6], D[6], d[6], xd[6], Ad[6];
    double S = lm_affine_residual(from, to, inl, m, h, A, v);
    for (int k = 0; k < 6; k++) D[k] = A[k * 6 + k];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1.0, lc = 0.75;
    for (int iter = 0; iter < max_iters; iter++) {
        double Al[36];
        memcpy(Al, A, sizeof(Al));
        for (int k = 0; k < 6; k++) Al[k * 6 + k] += lambda * D[k];
        if (!solve6(Al, v, d)) break;
        for (int k = 0; k < 6; k++) xd[k] = h[k] - d[k];
        const double Sd = lm_affine_residual(from, to, inl, m, xd, NULL, NULL);
        double dS = 0.0, t = 0.0, dmax = 0.0;
        for (int r = 0; r < 6; r++) {
            double s_ = 0.0;
            for (int c = 0; c < 6; c++) s_ += Al[r * 6 + c] * d[c];
            Ad[r] = 2.0 * v[r] - s_;
            dS += d[r] * Ad[r];
            t += d[r] * v[r];
            if (fabs(d[r]) > dmax) dmax = fabs(d[r]);
        }
        const double R = (S - Sd) / (fabs(dS) > 2.220446049250313e-16 ? dS : 1.0);
        if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0.0; }
        else if (R < Rlo) {
            double nu = (Sd - S) / (fabs(t) > 2.220446049250313e-16 ? t : 1.0) + 2.0;
            nu = nu < 2.0 ? 2.0 : (nu > 10.0 ? 10.0 : nu);
            if (lambda == 0.0) {
                double maxval = 2.220446049250313e-16;  /* 1 / max |diag(A^-1)|, via six solves */
                for (int k = 0; k < 6; k++) {
                    double e[6] = {0, 0, 0, 0, 0, 0}, col[6];
                    e[k] = 1.0;
                    if (solve6(A, e, col) && fabs(col[k]) > maxval) maxval = fabs(col[k]);
                }
                lambda = lc = 1.0 / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(h, xd, sizeof(xd));
            S = lm_affine_residual(from, to, inl, m, h, A, v);
        }
        if (!(dmax >= 1.1920929e-07 && S > 0.0)) break; /* epsx = epsf = FLT_EPSILON in createLMSolver(cb, maxIters) */
    }
}

/* from/to: m compacted pairs.  Out: M (2x3 row-major, float64), inl (m bytes).  Returns 1 on success. */
KO_API int ko_ransac_affine(const float *from, const float *to, int m, double Mout[6], uint8_t *inl, int *iters_used)
{
    for (int i = 0; i < m; i++) inl[i] = 0;
    if (iters_used) *iters_used = 0;
    if (m < 3) return 0;
    int *counts = (int *)malloc(sizeof(int) * RANSAC_MAX_ITERS);
#pragma omp parallel for schedule(dynamic, 8)
    for (int hyp = 0; hyp < RANSAC_MAX_ITERS; hyp++) {
        double M[6];
        int c = 0;
        if (hypothesis(from, to, m, (uint32_t)hyp, M))
            for (int i = 0; i < m; i++) c += is_inlier(M, from[2 * i], from[2 * i + 1], to[2 * i], to[2 * i + 1]);
        counts[hyp] = c;
    }
    int niters = RANSAC_MAX_ITERS, best = -1, best_count = 0, it;
    for (it = 0; it < niters; it++) {
        int c = counts[it];
        if (c > (best_count > 2 ? best_count : 2)) {
            best = it;
            best_count = c;
            niters = ransac_update_iters(RANSAC_CONF, (double)(m - c) / m, niters);
        }
    }
    free(counts);
    if (iters_used) *iters_used = it;
    if (best < 0) return 0;
    double M[6];
    hypothesis(from, to, m, (uint32_t)best, M);
    for (int i = 0; i < m; i++) inl[i] = (uint8_t)is_inlier(M, from[2 * i], from[2 * i + 1], to[2 * i], to[2 * i + 1]);

    /* least-squares refit on the inliers; every sum is an exact int64 fixed-point sum (order independent) */
    int64_t sx = 0, sy = 0, su = 0, sv = 0;
    for (int i = 0; i < m; i++)
        if (inl[i]) {
            sx += fixq(from[2 * i], 32); sy += fixq(from[2 * i + 1], 32);
            su += fixq(to[2 * i], 32);   sv += fixq(to[2 * i + 1], 32);
        }
    double cnt = (double)best_count;
    double mx = ldexp((double)sx, -32) / cnt, my = ldexp((double)sy, -32) / cnt;
    double mu = ldexp((double)su, -32) / cnt, mv = ldexp((double)sv, -32) / cnt;
    int64_t q[9] = {0};
    for (int i = 0; i < m; i++)
        if (inl[i]) {
            double x = from[2 * i] - mx, y = from[2 * i + 1] - my, u = to[2 * i] - mu, v = to[2 * i + 1] - mv;
            q[0] += fixq(x * x, 20); q[1] += fixq(x * y, 20); q[2] += fixq(y * y, 20);
            q[3] += fixq(x * u, 20); q[4] += fixq(y * u, 20);
            q[5] += fixq(x * v, 20); q[6] += fixq(y * v, 20);
        }
    double Sxx = ldexp((double)q[0], -20), Sxy = ldexp((double)q[1], -20), Syy = ldexp((double)q[2], -20);
    double Sxu = ldexp((double)q[3], -20), Syu = ldexp((double)q[4], -20), Sxv = ldexp((double)q[5], -20), Syv = ldexp((double)q[6], -20);
    double det = Sxx * Syy - Sxy * Sxy;
    if (best_count >= 3 && det > 1e-9 * (Sxx + Syy) * (Sxx + Syy) && det > 0) {
        double a = (Sxu * Syy - Syu * Sxy) / det, b = (Sxx * Syu - Sxy * Sxu) / det;
        double d = (Sxv * Syy - Syv * Sxy) / det, e = (Sxx * Syv - Sxy * Sxv) / det;
        M[0] = a; M[1] = b; M[2] = (mu - a * mx) - b * my;
        M[3] = d; M[4] = e; M[5] = (mv - d * mx) - e * my;
    }
    if (g_refine_mode == 1) {  /* sensitivity mode: OpenCV's own sequence -- best minimal-sample model + 10 LM iterations on its inliers */
        hypothesis(from, to, m, (uint32_t)best, M);
        if (best_count > 3) lm_refine_affine(from, to, inl, m, M, 10);
    }
    for (int k = 0; k < 6; k++) Mout[k] = M[k];
    return 1;
}

/* mean of (p - p0) over valid points, float32 differences summed exactly in 2^-32 fixed point (KLT.py:121-123) */
static void mean_translation(const float *p0, const float *p, const uint8_t *v, int n, double out[2], int *count)
{
    int64_t sx = 0, sy = 0;
    int c = 0;
    for (int i = 0; i < n; i++)
        if (v[i]) {
            float dx = p[2 * i] - p0[2 * i], dy = p[2 * i + 1] - p0[2 * i + 1];
            sx += fixq((double)dx, 32); sy += fixq((double)dy, 32);
            c++;
        }
    *count = c;
    out[0] = c ? ldexp((double)sx, -32) / (double)c : 0.0;
    out[1] = c ? ldexp((double)sy, -32) / (double)c : 0.0;
}

/* ------------------------------------------------------------------------------------------------ */
/* KLTregional / KLTmain                                                                             */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    int win, max_level, max_count;
    double eps;
} lk_params_t;

/* KLTregional (KLT.py:55-95).  T: 3x2 row-major float32.  p,v sized n. roi out = x0,x1,y0,y1 */
static void klt_regional(const uint8_t *im0, const uint8_t *im, int w, int h, int stride0, int stride, const float *p0, int n,
                         const float T[6], const lk_params_t *lk, float fbt, int translate, float *p, uint8_t *v, int roi[4],
                         uint8_t *warp_out)
{
    ko_bounding_rect(p0, n, w, h, 50, 50, roi);
    int x0 = roi[0], x1 = roi[1], y0 = roi[2], y1 = roi[3], rw = x1 - x0, rh = y1 - y0;
    float *p0r = (float *)malloc(sizeof(float) * 2 * n), *pa = (float *)malloc(sizeof(float) * 2 * n);
    float *err = (float *)malloc(sizeof(float) * n);
    float fx0 = (float)x0, fy0 = (float)y0;
    for (int i = 0; i < n; i++) { p0r[2 * i] = p0[2 * i] - fx0; p0r[2 * i + 1] = p0[2 * i + 1] - fy0; }
    uint8_t *warped = (uint8_t *)malloc((size_t)rw * rh);
    int dx = 0, dy = 0;
    if (translate) {
        dx = (int)T[4]; dy = (int)T[5]; /* truncation toward zero, KLT.py:66-67 */
        ko_crop_shift(im, w, h, stride, x0, x1, y0, y1, dx, dy, warped);
    } else {
        ko_remap_affine(im, w, h, stride, T, x0, x1, y0, y1, warped);
    }
    if (warp_out) memcpy(warp_out, warped, (size_t)rw * rh);
    ko_lk_fb(im0 + (ptrdiff_t)y0 * stride0 + x0, warped, rw, rh, stride0, rw, p0r, n, lk->win, lk->max_level, lk->max_count, lk->eps, fbt,
             pa, v, err, NULL);
    for (int i = 0; i < n; i++) {
        float ax = pa[2 * i] + fx0, ay = pa[2 * i + 1] + fy0;
        if (translate) { p[2 * i] = ax + (float)dx; p[2 * i + 1] = ay + (float)dy; }
        else { p[2 * i] = (ax * T[0] + ay * T[2]) + T[4]; p[2 * i + 1] = (ax * T[1] + ay * T[3]) + T[5]; }
    }
    free(p0r); free(pa); free(err); free(warped);
}

static int compact_pairs(const float *a, const float *b, const uint8_t *v, int n, float *ca, float *cb, int *idx)
{
    int m = 0;
    for (int i = 0; i < n; i++)
        if (v[i]) {
            ca[2 * m] = a[2 * i]; ca[2 * m + 1] = a[2 * i + 1];
            cb[2 * m] = b[2 * i]; cb[2 * m + 1] = b[2 * i + 1];
            idx[m++] = i;
        }
    return m;
}

typedef struct {        /* optional stage outputs for stage-by-stage parity tests (any pointer may be NULL) */
    float *p_small;     /* n x 2 : stage-1 LK result scaled back to full resolution */
    uint8_t *v_small;   /* n     : after the RANSAC inlier gate */
    double *T_trans;    /* 2     : mean translation */
    int *roi;           /* 4     : x0,x1,y0,y1 */
    float *p_coarse;    /* n x 2 : stage-2 result */
    uint8_t *v_coarse;  /* n */
    double *T23;        /* 6     : 2x3 affine */
    uint8_t *warped;    /* ROI-sized warp of stage 3 (caller allocates w*h) */
    float *p_fine;      /* n x 2 : stage-3 result for ALL points */
    int *flags;         /* 1     : bit0 = coarse-affine failure (KLT.py:128-130) */
} klt_stages_t;

/* KLTmain (KLT.py:99-134).  lk_*: coarse = (15, 4, 10, 0.1), fine = (51, 0, 30, 0.001) in the reference (:106-107).
 * im0_small may be NULL (computed then).  Outputs: p_all n x 2 (all points), v n, im_small (dw x dh). */
KO_API int ko_klt_main(const uint8_t *im, const uint8_t *im0, const uint8_t *im0_small, int w, int h, int stride, int stride0,
                       const float *p0, int n, int cw, int cl, int cc, double ce, int fw, int fl, int fc, double fe,
                       float *p_all, uint8_t *v, uint8_t *im_small, klt_stages_t *st)
{
    lk_params_t lkc = {cw, cl, cc, ce}, lkf = {fw, fl, fc, fe};
    int dw, dh, flags = 0;
    ko_resize_quarter_dims(w, h, &dw, &dh);
    ko_resize_quarter(im, w, h, stride, im_small);
    uint8_t *small0 = NULL;
    if (!im0_small) {
        small0 = (uint8_t *)malloc((size_t)dw * dh);
        ko_resize_quarter(im0, w, h, stride0, small0);
        im0_small = small0;
    }
    if (n <= 0) { /* no tracks left: nothing to track, v.sum() > 10 fails (KLT.py:126-130) -> the coarse-affine failure flag; empty ROI like the product's glue */
        free(small0);
        if (st && st->T_trans) { st->T_trans[0] = 0; st->T_trans[1] = 0; }
        if (st && st->roi) { st->roi[0] = 1; st->roi[1] = 1; st->roi[2] = 1; st->roi[3] = 1; }
        if (st && st->T23) { double I6[6] = {1, 0, 0, 0, 1, 0}; memcpy(st->T23, I6, sizeof(I6)); }
        if (st && st->flags) *st->flags = 1;
        return 1;
    }
    float *ps = (float *)malloc(sizeof(float) * 2 * n), *p = (float *)malloc(sizeof(float) * 2 * n);
    float *err = (float *)malloc(sizeof(float) * n);
    float *ca = (float *)malloc(sizeof(float) * 2 * n), *cb = (float *)malloc(sizeof(float) * 2 * n);
    int *idx = (int *)malloc(sizeof(int) * n);
    uint8_t *inl = (uint8_t *)malloc(n > 0 ? n : 1);

    /* 1. coarse LK on the quarter-scale frame (KLT.py:114-117) */
    for (int i = 0; i < 2 * n; i++) ps[i] = p0[i] * 0.25f;
    ko_lk_fb(im0_small, im_small, dw, dh, dw, dw, ps, n, lkc.win, lkc.max_level, lkc.max_count, lkc.eps, -1.f, p, v, err, NULL);
    for (int i = 0; i < 2 * n; i++) p[i] = p[i] / 0.25f;
    double M[6];
    int m = compact_pairs(p0, p, v, n, ca, cb, idx);
    if (ko_ransac_affine(ca, cb, m, M, inl, NULL)) {
        for (int k = 0; k < m; k++) v[idx[k]] = inl[k];
    } else {
        for (int i = 0; i < n; i++) v[i] = 0;
    }
    if (st && st->p_small) memcpy(st->p_small, p, sizeof(float) * 2 * n);
    if (st && st->v_small) memcpy(st->v_small, v, n);

    /* 2. translation-compensated coarse LK on the full-resolution ROI (KLT.py:121-124) */
    double tr[2];
    int cnt;
    mean_translation(p0, p, v, n, tr, &cnt);
    if (st && st->T_trans) { st->T_trans[0] = tr[0]; st->T_trans[1] = tr[1]; }
    float T[6] = {1.f, 0.f, 0.f, 1.f, (float)tr[0], (float)tr[1]};
    int roi[4];
    klt_regional(im0, im, w, h, stride0, stride, p0, n, T, &lkc, 1.0f, 1, p, v, roi, NULL);
    if (st && st->roi) memcpy(st->roi, roi, sizeof(roi));
    if (st && st->p_coarse) memcpy(st->p_coarse, p, sizeof(float) * 2 * n);
    if (st && st->v_coarse) memcpy(st->v_coarse, v, n);

    /* affine from the survivors (KLT.py:126-130); the SURF fallback is out of scope -> keep the translation */
    int nv = 0;
    for (int i = 0; i < n; i++) nv += v[i];
    int ok = 0;
    if (nv > 10) {
        m = compact_pairs(p0, p, v, n, ca, cb, idx);
        ok = ko_ransac_affine(ca, cb, m, M, inl, NULL);
    }
    if (!ok) {
        flags |= 1;
        M[0] = 1; M[1] = 0; M[2] = tr[0]; M[3] = 0; M[4] = 1; M[5] = tr[1];
    }
    if (st && st->T23) memcpy(st->T23, M, sizeof(M));

    /* 3. fine LK on the affine-warped ROI (KLT.py:133): T = T23.T as float32, 3x2 row-major */
    float Tf[6] = {(float)M[0], (float)M[3], (float)M[1], (float)M[4], (float)M[2], (float)M[5]};
    klt_regional(im0, im, w, h, stride0, stride, p0, n, Tf, &lkf, 0.3f, 0, p_all, v, roi, st ? st->warped : NULL);
    if (st && st->p_fine) memcpy(st->p_fine, p_all, sizeof(float) * 2 * n);
    if (st && st->flags) *st->flags = flags;

    free(ps); free(p); free(err); free(ca); free(cb); free(idx); free(inl); free(small0);
    return flags;
}

/* thin exported wrapper so tests can call KLTregional alone */
KO_API void ko_klt_regional(const uint8_t *im0, const uint8_t *im, int w, int h, int stride0, int stride, const float *p0, int n,
                            const float T[6], int win, int max_level, int max_count, double eps, float fbt, int translate, float *p,
                            uint8_t *v, int roi[4], uint8_t *warp_out)
{
    lk_params_t lk = {win, max_level, max_count, eps};
    klt_regional(im0, im, w, h, stride0, stride, p0, n, T, &lk, fbt, translate, p, v, roi, warp_out);
}

KO_API double ko_det_log(double x) { return det_log(x); }

/* cv2.cvtColor(BGR2GRAY) for 8-bit images (vidExample.py:91): OpenCV 4.x fixed point, 15 fractional bits (unpinned like the
 * rest of the OpenCV half) */
KO_API void ko_bgr2gray(const uint8_t *bgr, int w, int h, uint8_t *gray)
{
    for (size_t i = 0; i < (size_t)w * h; i++)
        gray[i] = (uint8_t)((bgr[3 * i] * 3735u + bgr[3 * i + 1] * 19235u + bgr[3 * i + 2] * 9798u + (1u << 14)) >> 15);
}


/* ------------------------------------------------------------------------------------------------ */
/* frame-0 initialisation: cv2.goodFeaturesToTrack(..., useHarrisDetector) */
/* and cv2.cornerSubPix, vidExample.py:110-115.  Restated from OpenCV 4.x's published algorithm;     */
/* PARITY UNPINNED like the rest of the OpenCV half.  Deliberate choice: the Harris structure tensor  */
/* is accumulated in exact integers (Sobel sums and 5x5 box sums of their products) and scaled once,  */
/* where OpenCV accumulates scaled floats in a build-dependent order.                                  */
/* ------------------------------------------------------------------------------------------------ */
static inline int sobel_dx(const uint8_t *im, int w, int h, ptrdiff_t st, int x, int y)
{
    int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w), ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    return (im[ym * st + xp] - im[ym * st + xm]) + 2 * (im[y * st + xp] - im[y * st + xm]) + (im[yp * st + xp] - im[yp * st + xm]);
}
static inline int sobel_dy(const uint8_t *im, int w, int h, ptrdiff_t st, int x, int y)
{
    int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w), ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    return (im[yp * st + xm] - im[ym * st + xm]) + 2 * (im[yp * st + x] - im[ym * st + x]) + (im[yp * st + xp] - im[ym * st + xp]);
}

/* Harris response (cornerHarris, blockSize x blockSize box, Sobel aperture 3, k): float32 [h][w] */
KO_API void ko_harris_response(const uint8_t *im, int w, int h, int stride, int block, double k, float *resp)
{
    int *dx = (int *)malloc(sizeof(int) * (size_t)w * h), *dy = (int *)malloc(sizeof(int) * (size_t)w * h);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            dx[(size_t)y * w + x] = sobel_dx(im, w, h, stride, x, y);
            dy[(size_t)y * w + x] = sobel_dy(im, w, h, stride, x, y);
        }
    const double scale = 1.0 / (4.0 * block * 255.0); /* (1 << (aperture-1)) * block * 255 */
    const float s2 = (float)(scale * scale), kf = (float)k;
    const int r0 = block / 2; /* anchor = block/2: window [x - r0, x - r0 + block) */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            long long sxx = 0, sxy = 0, syy = 0;
            for (int j = 0; j < block; j++) {
                int yy = reflect101(y - r0 + j, h);
                for (int i = 0; i < block; i++) {
                    int xx = reflect101(x - r0 + i, w);
                    int a = dx[(size_t)yy * w + xx], b = dy[(size_t)yy * w + xx];
                    sxx += a * a; sxy += a * b; syy += b * b;
                }
            }
            float a = (float)sxx * s2, b = (float)sxy * s2, c = (float)syy * s2;
            float tr = a + c;
            resp[(size_t)y * w + x] = (a * c - b * b) - (kf * tr) * tr;
        }
    free(dx); free(dy);
}

typedef struct { float v; int idx; } cand_t;
static int cand_cmp(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0); /* ties: higher address first (greaterThanPtr) */
}

/* goodFeaturesToTrack(image, maxCorners, quality, minDistance=0, blockSize, useHarris, k).  Returns the count; corners (x,y) float */
KO_API int ko_good_features(const uint8_t *im, int w, int h, int stride, int max_corners, double quality, int block, double k, float *corners)
{
    float *eig = (float *)malloc(sizeof(float) * (size_t)w * h);
    ko_harris_response(im, w, h, stride, block, k, eig);
    float mx = eig[0];
    for (size_t i = 1; i < (size_t)w * h; i++) if (eig[i] > mx) mx = eig[i];
    const float thr = (float)((double)mx * quality);
    /* threshold(TOZERO): keep values > thr, else 0 */
    for (size_t i = 0; i < (size_t)w * h; i++) if (!(eig[i] > thr)) eig[i] = 0.f;
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)w * h);
    int n = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float v = eig[(size_t)y * w + x];
            if (v == 0.f) continue;
            float m = v; /* 3x3 dilation */
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    float u = eig[(size_t)(y + j) * w + x + i];
                    if (u > m) m = u;
                }
            if (v == m) { c[n].v = v; c[n].idx = y * w + x; n++; }
        }
    qsort(c, n, sizeof(cand_t), cand_cmp);
    if (max_corners > 0 && n > max_corners) n = max_corners;
    for (int i = 0; i < n; i++) { corners[2 * i] = (float)(c[i].idx % w); corners[2 * i + 1] = (float)(c[i].idx / w); }
    free(c); free(eig);
    return n;
}

/* getRectSubPix(8u -> 32f) of a pw x ph patch centred at (cx, cy), replicated border */
static void rect_subpix(const uint8_t *im, int w, int h, ptrdiff_t st, float cx, float cy, int pw, int ph, float *dst)
{
    cx -= (pw - 1) * 0.5f; cy -= (ph - 1) * 0.5f;
    int ipx = ifloor(cx), ipy = ifloor(cy);
    float a = cx - ipx, b = cy - ipy;
    float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    for (int i = 0; i < ph; i++) {
        int y0 = ipy + i, y1 = y0 + 1;
        y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0);
        y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
        for (int j = 0; j < pw; j++) {
            int x0 = ipx + j, x1 = x0 + 1;
            x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0);
            x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
            float v = (float)im[y0 * st + x0] * a11;
            v = v + (float)im[y0 * st + x1] * a12;
            v = v + (float)im[y1 * st + x0] * a21;
            v = v + (float)im[y1 * st + x1] * a22;
            dst[i * pw + j] = v;
        }
    }
}

/* cornerSubPix(image, corners, winSize=(win,win), zeroZone=(-1,-1), criteria=(EPS+MAX_ITER, max_iter, eps)) in place */
KO_API void ko_corner_subpix(const uint8_t *im, int w, int h, int stride, float *pts, int n, int win, int max_iter, double eps)
{
    const int ww = 2 * win + 1, pw = ww + 2;
    if (max_iter < 1) max_iter = 1;
    if (max_iter > 100) max_iter = 100;
    if (eps < 0) eps = 0;
    eps *= eps;
    float *mask = (float *)malloc(sizeof(float) * ww * ww);
    for (int i = 0; i < ww; i++) {
        float y = (float)(i - win) / win, vy = expf(-y * y);
        for (int j = 0; j < ww; j++) {
            float x = (float)(j - win) / win;
            mask[i * ww + j] = (float)(vy * expf(-x * x));
        }
    }
#pragma omp parallel
    {
        float *buf = (float *)malloc(sizeof(float) * pw * pw);
#pragma omp for schedule(dynamic, 8)
        for (int q = 0; q < n; q++) {
            const float tx = pts[2 * q], ty = pts[2 * q + 1];
            float cx = tx, cy = ty;
            int iter = 0;
            double err = 0;
            do {
                double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
                rect_subpix(im, w, h, stride, cx, cy, pw, pw, buf);
                for (int i = 0; i < ww; i++) {
                    const float *sp = buf + (i + 1) * pw + 1;
                    const double py = i - win;
                    for (int j = 0; j < ww; j++) {
                        const double m = mask[i * ww + j];
                        const double tgx = sp[j + 1] - sp[j - 1], tgy = sp[j + pw] - sp[j - pw];
                        const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m, px = j - win;
                        a += gxx; b += gxy; c += gyy;
                        bb1 += gxx * px + gxy * py;
                        bb2 += gxy * px + gyy * py;
                    }
                }
                const double det = a * c - b * b;
                if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
                const double sc = 1.0 / det;
                const float nx = (float)(cx + c * sc * bb1 - b * sc * bb2);
                const float ny = (float)(cy - b * sc * bb1 + a * sc * bb2);
                err = (double)(nx - cx) * (nx - cx) + (double)(ny - cy) * (ny - cy);
                cx = nx; cy = ny;
                if (cx < 0 || cx >= w || cy < 0 || cy >= h) break;
            } while (++iter < max_iter && err > eps);
            if (fabsf(cx - tx) > win || fabsf(cy - ty) > win) { cx = tx; cy = ty; }
            pts[2 * q] = cx; pts[2 * q + 1] = cy;
        }
        free(buf);
    }
    free(mask);
}
