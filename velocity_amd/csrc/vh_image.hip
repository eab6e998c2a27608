// Image-stage kernels of the KLT path: quarter-scale nearest decimation (K1), pyrDown (K2), shifted crop (K8) and
// affine remap (K9).  All are HBM-bound byte kernels; descriptors are read from device memory because ROI sizes are
// data dependent (bounding box of the tracks) and the whole frame pipeline runs without a host round trip.
#include <algorithm>
#include "vh_kernels.hpp"
#include "vh_valu.hpp"

// descriptor b of a strided descriptor table (tables live inside per-stream structs)
__device__ __forceinline__ const ImgDesc& desc_at(const void* base, size_t stride, int b)
{
    return *reinterpret_cast<const ImgDesc*>(reinterpret_cast<const char*>(base) + (size_t)b * stride);
}

// ---------------------------------------------------------------------------------------------------------------
// cv2.resize(fx=fy=0.25, INTER_NEAREST)  (utils/KLT.py:111-113): dst[y,x] = src[min(4y,H-1), min(4x,W-1)]
// one thread = 4 consecutive output pixels (one packed dword store when the row is aligned)
// ---------------------------------------------------------------------------------------------------------------
// Descriptor tables: entry z belongs to stream z / per_stream and is its (z % per_stream)-th descriptor.
__global__ __launch_bounds__(256) void k_resize_quarter(const void* src_tab, const void* dst_tab, size_t tab_stride, int per_stream)
{
    const int b = blockIdx.z / per_stream, k = blockIdx.z - b * per_stream;
    const ImgDesc s = desc_at(reinterpret_cast<const ImgDesc*>(src_tab) + k, tab_stride, b);
    const ImgDesc d = desc_at(reinterpret_cast<const ImgDesc*>(dst_tab) + k, tab_stride, b);
    int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= d.h || x4 >= d.w) return;
    int sy = min(4 * y, s.h - 1);
    const uint8_t* srow = s.p + (size_t)sy * s.stride;
    uint8_t* drow = const_cast<uint8_t*>(d.p) + (size_t)y * d.stride;
    uint32_t pack = 0;
    int cnt = min(4, d.w - x4);
    const uint8_t* sp = srow + 4 * (size_t)x4;
    if (cnt == 4 && 4 * (x4 + 3) <= s.w - 1 && (reinterpret_cast<uintptr_t>(sp) & 3) == 0) {
        // the four samples are byte 0 of four consecutive dwords: one 16-byte load and two v_perm instead of four dependent byte loads
        const uint4 q = *reinterpret_cast<const uint4*>(sp);
        pack = __builtin_amdgcn_perm(q.y, q.x, 0x0c0c0400u) | __builtin_amdgcn_perm(q.w, q.z, 0x04000c0cu);
    } else {
        for (int k = 0; k < cnt; k++) pack |= (uint32_t)srow[min(4 * (x4 + k), s.w - 1)] << (8 * k);
    }
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(drow + x4) & 3) == 0)) {
        *reinterpret_cast<uint32_t*>(drow + x4) = pack;
    } else {
        for (int k = 0; k < cnt; k++) drow[x4 + k] = (uint8_t)(pack >> (8 * k));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pyrDown: dst = ((w+1)/2, (h+1)/2), separable [1 4 6 4 1], REFLECT_101, (sum + 128) >> 8   
// Streaming register kernel, no LDS: one thread owns 4 output columns and walks RB output rows downwards (RB = 8 when the launch fills the chip anyway, 2 for single-stream latency).  Per
// source row it loads 16 bytes (4 dwords, re-aligned with v_alignbyte), forms the 4 horizontal sums with
// v_dot4_u32_u8 (weights 1,4,6,4 + the fifth tap) and keeps the last five rows as packed uint16 pairs; the vertical pass
// is packed 16-bit math (the 5x5 sum is < 2^16).  Every source row is touched (2*RB+3)/(2*RB) times, adjacent
// lanes read adjacent 8-byte steps (coalesced, 2x overlap served by the cache).  Border threads take a byte path
// with REFLECT_101.
// Build tables: stream b = blockIdx.z / 2 owns two PyrBuild entries (previous / current image); level `lvl` of the
// pyramid is read, level lvl+1 is written; disabled entries and pyramids with fewer levels exit immediately.
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
typedef const unsigned __attribute__((address_space(1)))* pd_gptr;

struct PdRow { unsigned s0, s1, s2; };  // source bytes sx0 .. sx0+11 of one row

// byte path (REFLECT_101 per byte): tiny levels and the two rows where the dword path could leave the allocation
__device__ __forceinline__ PdRow pd_bytes_row(const ImgDesc& s, int sx0, int ry)
{
    const uint8_t* row = s.p + (size_t)ry * s.stride;
    unsigned v[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < 11; c++) v[c >> 2] |= (unsigned)row[vh_reflect101(sx0 + c, s.w)] << (8 * (c & 3));
    PdRow o;
    o.s0 = v[0]; o.s1 = v[1]; o.s2 = v[2];
    return o;
}

__device__ __forceinline__ unsigned pd_from_next_lane(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1 -> lane i reads lane i+1
}

__device__ __forceinline__ uint2 pd_hsum(const PdRow& r)
{
    const unsigned wgt = 0x04060401u;  // taps 1 4 6 4 on bytes 0..3, the fifth tap (x1) is added separately
    const unsigned h0 = __builtin_amdgcn_udot4(r.s0, wgt, r.s1 & 0xffu, false);
    const unsigned h1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(r.s1, r.s0, 2), wgt, (r.s1 >> 16) & 0xffu, false);
    const unsigned h2 = __builtin_amdgcn_udot4(r.s1, wgt, r.s2 & 0xffu, false);
    const unsigned h3 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(r.s2, r.s1, 2), wgt, (r.s2 >> 16) & 0xffu, false);
    return make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
}

__device__ __forceinline__ unsigned pd_vert(unsigned a, unsigned b, unsigned c, unsigned d, unsigned e)
{
    const ushort2v va = __builtin_bit_cast(ushort2v, a), vb = __builtin_bit_cast(ushort2v, b), vc = __builtin_bit_cast(ushort2v, c),
                   vd = __builtin_bit_cast(ushort2v, d), ve = __builtin_bit_cast(ushort2v, e);
    const ushort2v k4 = {4, 4}, k6 = {6, 6}, k128 = {128, 128}, k8 = {8, 8};
    const ushort2v acc = ((va + ve) + (vb + vd) * k4 + vc * k6 + k128) >> k8;  // <= 65408: no 16-bit overflow
    return __builtin_bit_cast(unsigned, acc);
}

// Lanes of a wave read adjacent 8-byte steps of a source row, and lane i needs bytes [8i-2, 8i+8]: it loads only its own
// 8 bytes (one dwordx2) and takes the next 8 from lane i+1 through DPP.  REFLECT_101 columns never cost a memory access:
// a lane's 12-byte stream already holds the mirrored pixels (left edge: columns 1,2 for -1,-2; right edge: columns w-2,
// w-3 for w, w+1), so three v_perm with per-lane selectors (identity for interior lanes) fix every row without a branch.
// All 2*RB+3 rows are requested before the first is used.
template <int RB>
__global__ __launch_bounds__(256) void k_pyr_down(const void* pb_tab, size_t ws_stride, int lvl)
{
    const PyrBuild& pb = reinterpret_cast<const PyrBuild*>(reinterpret_cast<const char*>(pb_tab) + (size_t)(blockIdx.z >> 1) * ws_stride)[blockIdx.z & 1];
    if (!pb.enable || pb.pyr == nullptr) return;
    const PyrDesc& P = *pb.pyr;
    if (lvl + 1 >= P.nlevels) return;
    const ImgDesc s = P.lv[lvl];
    const ImgDesc d = P.lv[lvl + 1];
    const int ox0 = (blockIdx.x * 64 + threadIdx.x) * 4, oy0 = (blockIdx.y * 4 + threadIdx.y) * RB;
    if (oy0 >= d.h) return;            // wave-uniform
    const bool live = ox0 < d.w;       // dead lanes stay for the DPP exchange, they load and store nothing
    const int sx0 = 2 * ox0 - 2, cnt = min(4, d.w - ox0);
    const bool wide = s.w >= 32;       // uniform per image; narrower levels go through the byte path
    const bool left = ox0 == 0, right = sx0 + 16 > s.w;     // right: the 16-byte read passes the end of the row
    const bool dwords = live && wide;
    // an interior lane always loads its own 8 bytes: only such a right neighbour can provide my second half
    const bool next_interior = threadIdx.x != 63 && ox0 + 4 < d.w && sx0 + 8 + 16 <= s.w;
    const bool own_ext = dwords && !next_interior;
    // per-lane byte selectors: stream position p takes position src(p) (mirror about column 0 / column w-1)
    unsigned sel0 = 0x03020100u, sel1 = 0x07060504u, sel2 = 0x07060504u;
    if (dwords && (left || right)) {
        const int m = s.w - 1 - sx0;  // stream position of the last column of the row
        unsigned sl[3] = {0, 0, 0};
        for (int p = 0; p < 12; p++) {
            int q = p;
            if (left && p < 2) q = 4 - p;
            if (q > m) q = 2 * m - q;
            const int rel = min(max(q - (p >= 8 ? 4 : 0), 0), 7);
            sl[p >> 2] |= (unsigned)rel << (8 * (p & 3));
        }
        sel0 = sl[0]; sel1 = sl[1]; sel2 = sl[2];
    }

    constexpr int NR = 2 * RB + 3;
    uint2 own[NR], ext[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int ry = vh_reflect101(2 * oy0 - 2 + k, s.h);
        const uintptr_t a = reinterpret_cast<uintptr_t>(s.p + (size_t)ry * s.stride + sx0);
        pd_gptr ap = (pd_gptr)(a - (a & 3));
        // the dword reads of an edge lane touch a few bytes of the neighbouring row: not before the first / after the last row
        const bool ok = dwords && !(left && ry == 0) && !(right && ry == s.h - 1);
        own[k] = make_uint2(0, 0); ext[k] = make_uint2(0, 0);
        if (ok) own[k] = make_uint2(ap[0], ap[1]);
        if (ok && own_ext) ext[k] = make_uint2(ap[2], ap[3]);
    }
    uint2 h[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int ry = vh_reflect101(2 * oy0 - 2 + k, s.h);
        const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(s.p + (size_t)ry * s.stride + sx0) & 3);
        const unsigned n0 = pd_from_next_lane(own[k].x), n1 = pd_from_next_lane(own[k].y);
        const unsigned d2 = own_ext ? ext[k].x : n0, d3 = own_ext ? ext[k].y : n1;
        const unsigned t0 = __builtin_amdgcn_alignbyte(own[k].y, own[k].x, sh), t1 = __builtin_amdgcn_alignbyte(d2, own[k].y, sh),
                       t2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
        PdRow r;
        r.s0 = __builtin_amdgcn_perm(t1, t0, sel0);
        r.s1 = __builtin_amdgcn_perm(t1, t0, sel1);
        r.s2 = __builtin_amdgcn_perm(t2, t1, sel2);
        if (live && (!wide || (left && ry == 0) || (right && ry == s.h - 1))) r = pd_bytes_row(s, sx0, ry);
        h[k] = pd_hsum(r);
    }
    if (!live) return;
#pragma unroll
    for (int r = 0; r < RB; r++) {
        const int oy = oy0 + r;
        const unsigned o01 = pd_vert(h[2 * r].x, h[2 * r + 1].x, h[2 * r + 2].x, h[2 * r + 3].x, h[2 * r + 4].x);
        const unsigned o23 = pd_vert(h[2 * r].y, h[2 * r + 1].y, h[2 * r + 2].y, h[2 * r + 3].y, h[2 * r + 4].y);
        const unsigned pack = __builtin_amdgcn_perm(o23, o01, 0x06040200u);
        if (oy < d.h) {
            uint8_t* dp = const_cast<uint8_t*>(d.p) + (size_t)oy * d.stride + ox0;
            if (cnt == 4 && (reinterpret_cast<uintptr_t>(dp) & 3) == 0) {
                *reinterpret_cast<uint32_t*>(dp) = pack;
            } else {
                for (int k = 0; k < cnt; k++) dp[k] = (uint8_t)(pack >> (8 * k));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ROI warp stage of KLTregional (utils/KLT.py:65-73).  mode 0: integer-shifted crop, zero outside the frame
// (translateFlag branch, SURVEY App. B intent); mode 1: float32 affine map + remap(INTER_LINEAR) with 5-bit
// fixed-point coordinates and 15-bit weights, constant-0 border.  One thread = 4 consecutive ROI pixels.
// ---------------------------------------------------------------------------------------------------------------
#define RW_ROWS 4  // ROI rows per thread: their source loads are all in flight together (one memory round trip per 16 pixels)

__device__ __forceinline__ void roi_store4(uint8_t* drow, int x4, int cnt, uint32_t pack)
{
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(drow + x4) & 3) == 0)) {
        *reinterpret_cast<uint32_t*>(drow + x4) = pack;
    } else {
        for (int k = 0; k < cnt; k++) drow[x4 + k] = (uint8_t)(pack >> (8 * k));
    }
}

// OpenCV's 15-bit weights w = 32 (32-ax | ax)(32-ay | ay) factor exactly: sum s w = 32 [32 t + ay (b - t)] with
// t = 32 s00 + ax (s01 - s00), b likewise -> (sum + 2^14) >> 15 == (32 t + ay (b - t) + 2^9) >> 10
__device__ __forceinline__ uint32_t remap_blend(int s00, int s01, int s10, int s11, int ax, int ay)
{
    const int t = 32 * s00 + ax * (s01 - s00), b = 32 * s10 + ax * (s11 - s10);
    return (uint32_t)((32 * t + ay * (b - t) + (1 << 9)) >> 10);
}

__global__ __launch_bounds__(256) void k_roi_warp(const void* job_tab, size_t tab_stride)
{
    const WarpJob J = *reinterpret_cast<const WarpJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blockIdx.z * tab_stride);
    if (J.mode < 0) return;
    const int rw = J.x1 - J.x0, rh = J.y1 - J.y0;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int ry0 = (blockIdx.y * blockDim.y + threadIdx.y) * RW_ROWS;
    if (ry0 >= rh || x4 >= rw) return;
    const ImgDesc s = J.src;
    const int cnt = min(4, rw - x4);
    if (J.mode == 0) {
        for (int r = 0; r < RW_ROWS && ry0 + r < rh; r++) {
            const int sy = J.y0 + ry0 + r + J.dy;
            const bool yin = sy >= 0 && sy < s.h;
            uint32_t pack = 0;
            for (int k = 0; k < cnt; k++) {
                int sx = J.x0 + x4 + k + J.dx;
                uint32_t v = (yin && sx >= 0 && sx < s.w) ? s.p[(size_t)sy * s.stride + sx] : 0u;
                pack |= v << (8 * k);
            }
            roi_store4(J.dst + (size_t)(ry0 + r) * J.dst_stride, x4, cnt, pack);
        }
        return;
    }
    // affine remap: coordinates of RW_ROWS x 4 pixels, then every row's source loads, then the blends
    float xa[4], xb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float x = (float)(J.x0 + x4 + k);
        xa[k] = __fmul_rn(x, J.T[0]);
        xb[k] = __fmul_rn(x, J.T[1]);
    }
    int fx[RW_ROWS][4], fy[RW_ROWS][4];
    unsigned t0[RW_ROWS], t1[RW_ROWS], b0[RW_ROWS], b1[RW_ROWS], sh0[RW_ROWS], sh1[RW_ROWS];
    // near-identity maps (the tracker's case): the 4 pixels of a row sample one source row pair at consecutive columns, so the 2 x 5 source
    // bytes come from two aligned dword pairs instead of 16 byte gathers.  The whole thread (4 rows) takes that path or none of it does:
    // one branch per thread, straight-line code inside (per-row branches cost a third of the kernel's instructions in exec juggling).
    unsigned bad = (cnt != 4 || ry0 + RW_ROWS > rh) ? ~0u : 0u;
    const unsigned bsh = (unsigned)(reinterpret_cast<uintptr_t>(s.p) & 3);
    const uint8_t* bp = s.p - bsh;  // dword aligned, wave uniform
#pragma unroll
    for (int r = 0; r < RW_ROWS; r++) {
        const float y = (float)(J.y0 + ry0 + r);
        const float yx = __fmul_rn(y, J.T[2]), yy = __fmul_rn(y, J.T[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // float32, one rounding per operation, no fma (numpy: x*T00 + y*T10 + T20)
            const float mx = __fadd_rn(__fadd_rn(xa[k], yx), J.T[4]);
            const float my = __fadd_rn(__fadd_rn(xb[k], yy), J.T[5]);
            fx[r][k] = vh_round(__fmul_rn(mx, 32.f));
            fy[r][k] = vh_round(__fmul_rn(my, 32.f));
        }
        const int sx0 = fx[r][0] >> 5, sy0 = fy[r][0] >> 5;
        // (fx[k] >> 5) == sx0 + k  <=>  0 <= fx[k] - 32 (sx0 + k) < 32;  (fy[k] >> 5) == sy0  <=>  (fy[k] ^ fy[0]) < 32: one unsigned compare of the OR;
        // source window inside the frame: sx0 >= 3, sx0 + 8 <= w, 0 <= sy0 < h - 1 as unsigned range tests
        const int bx = fx[r][0] & ~31;
        const unsigned spread = (unsigned)(fx[r][1] - bx - 32) | (unsigned)(fx[r][2] - bx - 64) | (unsigned)(fx[r][3] - bx - 96) |
                                (unsigned)(fy[r][1] ^ fy[r][0]) | (unsigned)(fy[r][2] ^ fy[r][0]) | (unsigned)(fy[r][3] ^ fy[r][0]);
        const bool ok = spread < 32u && (unsigned)(sx0 - 3) <= (unsigned)(s.w - 11) && (unsigned)sy0 < (unsigned)(s.h - 1) && s.w >= 11;
        bad |= ok ? 0u : ~0u;
        // unconditional loads (a branch around a load makes the compiler wait per row): rows that do not qualify read the first bytes of the frame
        const unsigned o0 = (ok ? (unsigned)(__mul24(sy0, s.stride) + sx0) : 0u) + bsh, o1 = o0 + (ok ? (unsigned)s.stride : 0u);
        sh0[r] = o0 & 3u; sh1[r] = o1 & 3u;
        pd_gptr p0 = (pd_gptr)(bp + (o0 & ~3u)), p1 = (pd_gptr)(bp + (o1 & ~3u));
        t0[r] = p0[0]; t1[r] = p0[1]; b0[r] = p1[0]; b1[r] = p1[1];
    }
    if (((reinterpret_cast<uintptr_t>(J.dst) | (uintptr_t)J.dst_stride) & 3) != 0) bad = ~0u;  // packed dword stores need dword rows (x4 is a multiple of 4)
    if (bad == 0u) {
#pragma unroll
        for (int r = 0; r < RW_ROWS; r++) {
            // remap_blend on packed int16 pairs: t = (32-ax) s00 + ax s01 and b likewise are two v_dot2 of the byte pairs (k, k+1) with
            // (32-ax | ax << 16) = 65535 ax + 32; the result (32-ay) t + ay b + 2^9 is a third one (t, b <= 8160).  Same integers.
            const unsigned tl = __builtin_amdgcn_alignbyte(t1[r], t0[r], sh0[r]), th = t1[r] >> (8 * sh0[r]);
            const unsigned bl = __builtin_amdgcn_alignbyte(b1[r], b0[r], sh1[r]), bh = b1[r] >> (8 * sh1[r]);
            unsigned tp[4], bq[4];
            tp[0] = __builtin_amdgcn_perm(0u, tl, 0x0c010c00u); bq[0] = __builtin_amdgcn_perm(0u, bl, 0x0c010c00u);
            tp[1] = __builtin_amdgcn_perm(0u, tl, 0x0c020c01u); bq[1] = __builtin_amdgcn_perm(0u, bl, 0x0c020c01u);
            tp[2] = __builtin_amdgcn_perm(0u, tl, 0x0c030c02u); bq[2] = __builtin_amdgcn_perm(0u, bl, 0x0c030c02u);
            tp[3] = __builtin_amdgcn_perm(th, tl, 0x0c040c03u); bq[3] = __builtin_amdgcn_perm(bh, bl, 0x0c040c03u);
            uint32_t pack = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned wx = __umul24((unsigned)(fx[r][k] & 31), 65535u) + 32u, wy = __umul24((unsigned)(fy[r][k] & 31), 65535u) + 32u;
                const unsigned tb = (unsigned)dot2_first(tp[k], wx) | ((unsigned)dot2_first(bq[k], wx) << 16);
                pack |= (uint32_t)(dot2_s(tb, wy, 1 << 9) >> 10) << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(J.dst + (size_t)(ry0 + r) * J.dst_stride + x4) = pack;
        }
        return;
    }
    for (int r = 0; r < RW_ROWS && ry0 + r < rh; r++) {
        uint32_t pack = 0;
        for (int k = 0; k < cnt; k++) {
            const int sx = fx[r][k] >> 5, sy = fy[r][k] >> 5;
            const bool x0in = sx >= 0 && sx < s.w, x1in = sx + 1 >= 0 && sx + 1 < s.w;
            const bool y0in = sy >= 0 && sy < s.h, y1in = sy + 1 >= 0 && sy + 1 < s.h;
            const uint8_t* r0 = s.p + (ptrdiff_t)sy * s.stride + sx;
            const int s00 = (y0in && x0in) ? r0[0] : 0, s01 = (y0in && x1in) ? r0[1] : 0;
            const int s10 = (y1in && x0in) ? r0[s.stride] : 0, s11 = (y1in && x1in) ? r0[s.stride + 1] : 0;
            pack |= remap_blend(s00, s01, s10, s11, fx[r][k] & 31, fy[r][k] & 31) << (8 * k);
        }
        roi_store4(J.dst + (size_t)(ry0 + r) * J.dst_stride, x4, cnt, pack);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// frame ingest: cv2.cvtColor(imbgr, COLOR_BGR2GRAY), vidExample.py:91 -- OpenCV 4.x 8-bit path:
// gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15.  One thread = 4 pixels (12 bytes in, one packed dword out).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t* bgr, int w, int h, size_t sstride, uint8_t* gray, size_t dstride)
{
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x4 >= w) return;
    const uint8_t* s = bgr + (size_t)y * sstride + 3 * (size_t)x4;
    uint8_t* d = gray + (size_t)y * dstride + x4;
    const int cnt = min(4, w - x4);
    uint32_t pack = 0;
    for (int k = 0; k < cnt; k++) {
        const uint32_t v = (s[3 * k] * 3735u + s[3 * k + 1] * 19235u + s[3 * k + 2] * 9798u + (1u << 14)) >> 15;
        pack |= v << (8 * k);
    }
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) *reinterpret_cast<uint32_t*>(d) = pack;
    else for (int k = 0; k < cnt; k++) d[k] = (uint8_t)(pack >> (8 * k));
}

void vh_launch_bgr2gray(const uint8_t* bgr, int w, int h, size_t sstride, uint8_t* gray, size_t dstride, hipStream_t s)
{
    dim3 blk(64, 4), grd((w + 255) / 256, (h + 3) / 4);
    hipLaunchKernelGGL(k_bgr2gray, grd, blk, 0, s, bgr, w, h, sstride, gray, dstride);
}

// ---------------------------------------------------------------------------------------------------------------
// frame ingest, optional rescale: cv2.resize(im, (0,0), fx=scale, fy=scale, INTER_NEAREST) (vidExample.py:99-102).
// dst[y][x] = src[min(floor(y / fy), h-1)][min(floor(x / fx), w-1)], one thread = 4 output pixels
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_nearest(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, int dw, int dh, size_t dstride,
                                                        double ifx, double ify)
{
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= dh || x4 >= dw) return;
    const int sy = min((int)floor(y * ify), h - 1);
    const uint8_t* srow = src + (size_t)sy * sstride;
    uint8_t* d = dst + (size_t)y * dstride + x4;
    const int cnt = min(4, dw - x4);
    uint32_t pack = 0;
    for (int k = 0; k < cnt; k++) pack |= (uint32_t)srow[min((int)floor((x4 + k) * ifx), w - 1)] << (8 * k);
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) *reinterpret_cast<uint32_t*>(d) = pack;
    else for (int k = 0; k < cnt; k++) d[k] = (uint8_t)(pack >> (8 * k));
}

void vh_launch_resize_nearest(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, int dw, int dh, size_t dstride, double ifx, double ify,
                              hipStream_t s)
{
    dim3 blk(64, 4), grd((dw + 255) / 256, (dh + 3) / 4);
    hipLaunchKernelGGL(k_resize_nearest, grd, blk, 0, s, src, w, h, sstride, dst, dw, dh, dstride, ifx, ify);
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------------------
void vh_launch_resize_quarter(const void* src_tab, const void* dst_tab, size_t tab_stride, int per_stream, int batch, int max_dw, int max_dh,
                              hipStream_t s)
{
    dim3 blk(64, 4), grd((max_dw + 255) / 256, (max_dh + 3) / 4, batch * per_stream);
    hipLaunchKernelGGL(k_resize_quarter, grd, blk, 0, s, src_tab, dst_tab, tab_stride, per_stream);
}

// REFLECT_101 border ring of a freshly built small level (see VH_LV_PAD): one thread per dword of 4 ring pixels (the interior origin and the row
// pitch are dword aligned; a dword that straddles the image edge re-writes interior pixels with themselves).  Same build table as k_pyr_down;
// levels without a border (large levels, user buffers) and missing levels exit at once.
__global__ __launch_bounds__(256) void k_pyr_pad(const void* pb_tab, size_t ws_stride, int lvl)
{
    const PyrBuild& pb = reinterpret_cast<const PyrBuild*>(reinterpret_cast<const char*>(pb_tab) + (size_t)(blockIdx.z >> 1) * ws_stride)[blockIdx.z & 1];
    if (!pb.enable || pb.pyr == nullptr) return;
    const PyrDesc& P = *pb.pyr;
    if (lvl + 1 >= P.nlevels) return;
    const ImgDesc d = P.lv[lvl + 1];
    const int B = d.pad;
    if (B <= 0) return;
    const int wq = (d.w + 2 * B + 3) >> 2;        // dwords of a full-width ring row (columns -B .. w+B-1, rounded up: the pitch has 4 spare bytes)
    const int rq = ((d.w & 3) + B + 3) >> 2;       // dwords of the right band of an image row, starting at column w & ~3
    const int band = wq * 2 * B, side = (B / 4) + rq;
    const int t = blockIdx.x * 256 + threadIdx.x;
    int x, y;
    if (t < band) {  // the 2 B full-width rows above and below
        const int q = t / wq;
        x = 4 * (t - q * wq) - B;
        y = q < B ? q - B : d.h + q - B;
    } else {         // left (B / 4 dwords) and right (rq dwords) of the image rows
        const int u = t - band;
        if (u >= d.h * side) return;
        y = u / side;
        const int q = u - y * side;
        x = q < B / 4 ? 4 * q - B : (d.w & ~3) + 4 * (q - B / 4);
    }
    const uint8_t* srow = d.p + (ptrdiff_t)vh_reflect101(y, d.h) * d.stride;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (uint32_t)srow[vh_reflect101(x + k, d.w)] << (8 * k);
    *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(d.p) + (ptrdiff_t)y * d.stride + x) = v;
}

void vh_launch_pyr_down_ws(const void* pb_tab, size_t ws_stride, int batch, int lvl, int max_w0, int max_h0, hipStream_t s)
{
    // dims of level lvl+1 when level 0 is max_w0 x max_h0
    int w = max_w0, h = max_h0;
    for (int l = 0; l <= lvl; l++) { w = (w + 1) / 2; h = (h + 1) / 2; }
    dim3 blk(64, 4);
    // 4 output rows per thread (11 source rows in flight) once the launch fills the chip; 2 for single-stream latency
    if ((long long)w * h * batch >= (1ll << 20)) {
        dim3 grd((w + 255) / 256, (h + 15) / 16, batch * 2);
        hipLaunchKernelGGL(k_pyr_down<4>, grd, blk, 0, s, pb_tab, ws_stride, lvl);
    } else {
        dim3 grd((w + 255) / 256, (h + 7) / 8, batch * 2);
        hipLaunchKernelGGL(k_pyr_down<2>, grd, blk, 0, s, pb_tab, ws_stride, lvl);
    }
    // border ring of the new level when it is a small one (decided per image on the device).  The launch covers the worst small level inside w x h:
    // the ring dword count grows linearly with each dimension, so over {w' <= w, h' <= h, w' h' <= VH_LV_PAD_MAX_PIXELS} it peaks in a corner --
    // the widest level with the rows that still fit (e.g. 8192 x 8) or the tallest one
    {
        auto ring_dwords = [](int ww, int hh) { return ((ww + 2 * VH_LV_PAD + 3) / 4) * 2 * VH_LV_PAD + hh * (VH_LV_PAD / 4 + (VH_LV_PAD + 6) / 4); };
        const int wa = std::min(w, VH_LV_PAD_MAX_PIXELS), ha = std::min(h, std::max(1, VH_LV_PAD_MAX_PIXELS / wa));
        const int hb = std::min(h, VH_LV_PAD_MAX_PIXELS), wb = std::min(w, std::max(1, VH_LV_PAD_MAX_PIXELS / hb));
        const int ring = std::max(ring_dwords(wa, ha), ring_dwords(wb, hb));
        hipLaunchKernelGGL(k_pyr_pad, dim3((ring + 255) / 256, 1, batch * 2), dim3(256), 0, s, pb_tab, ws_stride, lvl);
    }
}

void vh_launch_roi_warp(const void* job_tab, size_t tab_stride, int batch, int max_w, int max_h, hipStream_t s)
{
    dim3 blk(64, 4), grd((max_w + 255) / 256, (max_h + 4 * RW_ROWS - 1) / (4 * RW_ROWS), batch);
    hipLaunchKernelGGL(k_roi_warp, grd, blk, 0, s, job_tab, tab_stride);
}
