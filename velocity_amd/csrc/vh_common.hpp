// Shared device/host definitions for libvelocity_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VH_MAX_LEVELS 8
#define VH_WAVE 64

#define VH_CHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            vh_set_error(#expr, _e, __FILE__, __LINE__);                                       \
            return (int)_e ? (int)_e : -1;                                                     \
        }                                                                                      \
    } while (0)

void vh_set_error(const char* what, hipError_t e, const char* file, int line);

// ---- image / pyramid descriptors (device resident; dims may be data dependent) -------------------------------
// SMALL pyramid levels >= 1 that the library allocates itself (w * h <= VH_LV_PAD_MAX_PIXELS: there a large share of the windows touches the
// border) carry a REFLECT_101 border of VH_LV_PAD pixels on every side (rows VH_LV_STRIDE(w) bytes apart, >= 4 spare bytes per row), filled by
// k_pyr_pad right after the level is built: the row loads of windows at the border of such a level are then plain dword loads.  On large levels
// border windows are rare and a border ring would cost more to write than it saves.
#define VH_LK_STAT_SLOTS 64  // LKJob::stats points at [VH_LK_STAT_SLOTS][16] counters: a workgroup uses slot (launch slot & 63), entries 0 / 1 (see StreamWS::lk_stats)
#define VH_LV_PAD 24
#define VH_LV_PAD_MAX_PIXELS 65536
#define VH_LV_STRIDE(w) (((((w) + 2 * VH_LV_PAD) + 3) & ~3) + 4)
#define VH_LV_PADDED(w, h) ((long long)(w) * (h) <= VH_LV_PAD_MAX_PIXELS)

struct ImgDesc {
    const uint8_t* p;  // pixel (x,y) at p[y*stride + x]
    int w, h, stride;
    int pad;  // pixels of valid REFLECT_101 border around the w x h image (0: none)
};

struct PyrDesc {
    int nlevels;  // levels actually present (OpenCV truncation rule applied)
    int pad;
    ImgDesc lv[VH_MAX_LEVELS];
};

// how LK maps its result back to frame coordinates (KLT.py:86-89,114-115)
enum { VH_OUT_SCALE = 0, VH_OUT_TRANSLATE = 1, VH_OUT_AFFINE = 2 };

// One pyramidal-LK call (cv2calcOpticalFlowPyrLK, KLT.py:37-51) for one video stream.
struct LKJob {
    PyrDesc I, J;        // previous / next image pyramids
    const float* p_in;   // n x 2 points in frame coordinates
    float* p_out;        // n x 2 mapped-back result
    uint8_t* v_out;      // n     status (after the forward-backward gate when fbt >= 0)
    float* err_out;      // n     LK err of the forward pass (may be null)
    float* fbe_out;      // n     forward-backward error (may be null)
    float* praw_out;     // n x 2 un-mapped forward result in LK image coordinates (may be null)
    const int* n_ptr;    // device count of points (null -> n)
    const int* order;    // may be null: LAUNCH order of the points (a permutation of 0..n-1: workgroup slot k solves point order[k]); results stay where they were
    unsigned long long* stats;  // may be null: [0] += Newton iterations, [1] += template set-ups (profiling aid)
    int n;
    int win, max_level, max_count;
    double eps2;         // criteria epsilon, already clamped and squared
    float fbt;           // < 0: no backward pass
    float in_scale;      // p = p_in * in_scale - in_off   (in float32, as the reference does)
    float in_off[2];
    int out_mode;
    float out_scale;     // VH_OUT_SCALE:     p_out = p / out_scale
    float out_off[2];    // VH_OUT_TRANSLATE: p_out = (p + in_off) + out_off ; VH_OUT_AFFINE: [p + in_off, 1] @ T
    float T[6];          // 3x2 row-major float32
};

// ---- device helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ int vh_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
// REFLECT_101, branch free for |i| <= 4 (n - 1): the reflection has period P = 2 (n - 1) and on [0, P] it is min(a, P - a); one round of
// r = min(|i|, P - |i|) lands in range for |i| <= P, a second one for |i| <= 2 P.  The LK windows stay within a level's width plus one and a
// half windows, so on every level wider than ~ win / 2 their columns / rows satisfy the bound and the (never taken, wave-uniform) branch below
// costs two instructions.  Level 0 of an image or ROI SMALLER than that (the API only asks for 4 x 4; OpenCV's level-truncation rule does not
// apply to level 0) takes the looped form.  Using the looped form everywhere cost a divergent loop PER BYTE in a border load, i.e. one memory
// round trip per byte instead of one per row.
__device__ __forceinline__ int vh_reflect101_near(int i, int n)
{
    const int P = 2 * (n - 1);
    int a = max(i, -i);
    if (__builtin_expect(a > 2 * P, 0)) return vh_reflect101(i, n);
    int r = min(a, P - a);
    a = max(r, -r);
    r = min(a, P - a);
    return n == 1 ? 0 : r;
}
// correctly rounded float32 sqrt (HIP's __fsqrt_rn is the approximate native instruction; sqrtf is IEEE under the
// default -fhip-fp32-correctly-rounded-divide-sqrt)
__device__ __forceinline__ float vh_sqrtf(float v) { return __builtin_sqrtf(v); }
__device__ __forceinline__ int vh_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ int vh_round(float v) { return __float2int_rn(v); }  // round-half-even (cvRound)
__device__ __forceinline__ int vh_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// Exact integer sums over the 64 lanes, uniform result.  DPP row reductions (quad_perm, quad_perm, row_half_mirror, row_mirror: after four adds every lane
// holds its 16-lane row's total) + one v_readlane per row: ~25 / ~45 instructions and no LDS round trip, where the six-step __shfl_xor butterfly of rounds
// 1-5 cost 6 (12 for int64) dependent ds_bpermute round trips -- 3.9 us for the seven sums of the RANSAC refit in a single-stream launch (round 6, in-kernel
// stamps).  Integer addition is associative: bit-identical to any other order.
template <int CTRL>
__device__ __forceinline__ int vh_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ long long vh_wave_sum_i64(long long v)
{
#define VH_I64_DPP_STEP(ctrl)                                                                                                     \
    {                                                                                                                             \
        const unsigned lo = (unsigned)vh_dpp_i32<ctrl>((int)(unsigned)(unsigned long long)v);                                     \
        const unsigned hi = (unsigned)vh_dpp_i32<ctrl>((int)(unsigned)((unsigned long long)v >> 32));                             \
        v += (long long)(((unsigned long long)hi << 32) | lo);                                                                    \
    }
    VH_I64_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]
    VH_I64_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]
    VH_I64_DPP_STEP(0x141)  // row_half_mirror
    VH_I64_DPP_STEP(0x140)  // row_mirror
#undef VH_I64_DPP_STEP
    long long t = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, 16 * r);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), 16 * r);
        t += (long long)(((unsigned long long)hi << 32) | lo);
    }
    return t;
}
__device__ __forceinline__ double vh_wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int vh_wave_sum_i32(int v)
{
    v += vh_dpp_i32<0xB1>(v);
    v += vh_dpp_i32<0x4E>(v);
    v += vh_dpp_i32<0x141>(v);
    v += vh_dpp_i32<0x140>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// fixed-point conversion used by every order-independent reduction: round-half-even of v * 2^bits
__device__ __forceinline__ long long vh_fixq(double v, int bits) { return __double2ll_rn(ldexp(v, bits)); }

// stream-ordered upload of a small POD (<= 3 KiB) without hipMemcpy: the blob travels as a kernel argument
template <typename T>
__global__ void vh_k_store(T* dst, T value)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = value;
}
template <typename T>
static inline hipError_t vh_store(T* dst, const T& value, hipStream_t s)
{
    hipLaunchKernelGGL(vh_k_store<T>, dim3(1), dim3(64), 0, s, dst, value);
    return hipGetLastError();
}
