// Image-stage kernels of the KLT path: quarter-scale nearest decimation (K1), pyrDown (K2), shifted crop (K8) and
// affine remap (K9).  All are HBM-bound byte kernels; descriptors are read from device memory because ROI sizes are
// data dependent (bounding box of the tracks) and the whole frame pipeline runs without a host round trip.
#include <atomic>
#include <cstdlib>
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
#define RQ_ROWS 4  // output rows per thread: a thread's whole work used to be one 16-byte load and one dword store -- workgroups that live ~1 us
__global__ __launch_bounds__(256) void k_resize_quarter(const void* src_tab, const void* dst_tab, size_t tab_stride, int per_stream)
{
    const int b = blockIdx.z / per_stream, k = blockIdx.z - b * per_stream;
    const ImgDesc s = desc_at(reinterpret_cast<const ImgDesc*>(src_tab) + k, tab_stride, b);
    const ImgDesc d = desc_at(reinterpret_cast<const ImgDesc*>(dst_tab) + k, tab_stride, b);
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y0 = (blockIdx.y * blockDim.y + threadIdx.y) * RQ_ROWS;
    if (y0 >= d.h || x4 >= d.w) return;
    const int cnt = min(4, d.w - x4);
    const bool quad = cnt == 4 && 4 * (x4 + 3) <= s.w - 1 && ((reinterpret_cast<uintptr_t>(s.p) | (uintptr_t)s.stride) & 3) == 0;  // (x4 is a multiple of 4: 16-byte steps on dword rows)
    uint32_t pack[RQ_ROWS];
    if (quad) {
        // the four samples are byte 0 of four consecutive dwords: one 16-byte load and two v_perm instead of four dependent byte loads; all rows' loads first
        uint4 q[RQ_ROWS];
#pragma unroll
        for (int r = 0; r < RQ_ROWS; r++) {
            const int sy = min(4 * min(y0 + r, d.h - 1), s.h - 1);
            q[r] = *reinterpret_cast<const uint4*>(s.p + (size_t)sy * s.stride + 4 * (size_t)x4);
        }
#pragma unroll
        for (int r = 0; r < RQ_ROWS; r++) pack[r] = __builtin_amdgcn_perm(q[r].y, q[r].x, 0x0c0c0400u) | __builtin_amdgcn_perm(q[r].w, q[r].z, 0x04000c0cu);
    } else {
#pragma unroll
        for (int r = 0; r < RQ_ROWS; r++) {
            const int sy = min(4 * min(y0 + r, d.h - 1), s.h - 1);
            const uint8_t* srow = s.p + (size_t)sy * s.stride;
            pack[r] = 0;
            for (int k = 0; k < cnt; k++) pack[r] |= (uint32_t)srow[min(4 * (x4 + k), s.w - 1)] << (8 * k);
        }
    }
#pragma unroll
    for (int r = 0; r < RQ_ROWS; r++) {
        if (y0 + r >= d.h) break;
        uint8_t* drow = const_cast<uint8_t*>(d.p) + (size_t)(y0 + r) * d.stride;
        if (cnt == 4 && ((reinterpret_cast<uintptr_t>(drow + x4) & 3) == 0)) {
            *reinterpret_cast<uint32_t*>(drow + x4) = pack[r];
        } else {
            for (int k = 0; k < cnt; k++) drow[x4 + k] = (uint8_t)(pack[r] >> (8 * k));
        }
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
typedef const PyrDesc __attribute__((address_space(1)))* pd_gdesc;
__device__ __forceinline__ ImgDesc pd_level(pd_gdesc P, int l)
{
    ImgDesc d;
    d.p = P->lv[l].p; d.w = P->lv[l].w; d.h = P->lv[l].h; d.stride = P->lv[l].stride; d.pad = P->lv[l].pad;
    return d;
}

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
    const unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const PyrBuild& pb = reinterpret_cast<const PyrBuild*>(reinterpret_cast<const char*>(pb_tab) + (size_t)(bz >> 1) * ws_stride)[bz & 1];
    if (!pb.enable || pb.pyr == nullptr) return;
    // the pyramid descriptor hangs off a pointer that was itself loaded: as a GLOBAL pointer its fields come through the scalar cache (a generic one made
    // every wavefront start with a chain of three dependent flat loads before its first image row was requested -- of a workgroup that lives ~6 us)
    const pd_gdesc Pg = (pd_gdesc)pb.pyr;
    if (lvl + 1 >= Pg->nlevels) return;
    const ImgDesc s = pd_level(Pg, lvl);
    const ImgDesc d = pd_level(Pg, lvl + 1);
    const int ox0 = (int)(bx * blockDim.x + threadIdx.x) * 4, oy0 = (int)(by * blockDim.y + threadIdx.y) * RB;
    if (oy0 >= d.h) return;            // wave-uniform
    const bool live = ox0 < d.w;       // dead lanes stay for the DPP exchange, they load and store nothing
    const int sx0 = 2 * ox0 - 2, cnt = min(4, d.w - ox0);
    const bool wide = s.w >= 32;       // uniform per image; narrower levels go through the byte path
    const bool left = ox0 == 0, right = sx0 + 16 > s.w;     // right: the 16-byte read passes the end of the row
    const bool dwords = live && wide;
    // an interior lane always loads its own 8 bytes: only such a right neighbour can provide my second half
    const bool next_interior = threadIdx.x != blockDim.x - 1 && ox0 + 4 < d.w && sx0 + 8 + 16 <= s.w;  // (lane + 1 is my right neighbour only inside a block row)
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
    // ROW-INTERIOR WAVEFRONT (all but the first and last row blocks of a level): none of the 2 RB + 3 source rows is mirrored or is the first / last
    // row of the image, so no lane ever needs the byte path -- the left / right edge lanes are fixed by their byte selectors alone, lanes right of
    // the image load a clamped (in-row) address and store nothing.  A wave-uniform test sends the wavefront down a path WITHOUT per-lane branches:
    // every lane loads its own 16 aligned bytes per row (the second half overlaps the right neighbour's: served by the cache), 3 v_alignbyte +
    // 3 v_perm + 4 v_dot4 per row.  The general path below spends most of its instructions on exec-mask bookkeeping around byte paths it never takes.
    if (wide && 2 * oy0 - 2 >= 1 && 2 * oy0 - 2 + NR <= s.h - 1 && oy0 + RB <= d.h) {
        const int sxl = live ? sx0 : 0;  // dead lanes: any in-row address
        uint4 q[NR];
        unsigned shq[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(s.p + (size_t)(2 * oy0 - 2 + k) * s.stride + sxl);
            pd_gptr ap = (pd_gptr)(a - (a & 3));
            shq[k] = (unsigned)(a & 3);
            q[k] = make_uint4(ap[0], ap[1], ap[2], ap[3]);
        }
        uint2 hq[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const unsigned t0 = __builtin_amdgcn_alignbyte(q[k].y, q[k].x, shq[k]), t1 = __builtin_amdgcn_alignbyte(q[k].z, q[k].y, shq[k]),
                           t2 = __builtin_amdgcn_alignbyte(q[k].w, q[k].z, shq[k]);
            PdRow r;
            r.s0 = __builtin_amdgcn_perm(t1, t0, sel0);
            r.s1 = __builtin_amdgcn_perm(t1, t0, sel1);
            r.s2 = __builtin_amdgcn_perm(t2, t1, sel2);
            hq[k] = pd_hsum(r);
        }
        if (!live) return;
        const bool dw_ok = cnt == 4 && ((reinterpret_cast<uintptr_t>(d.p) | (uintptr_t)d.stride) & 3) == 0;  // ox0 is a multiple of 4: dword stores on dword rows
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const unsigned o01 = pd_vert(hq[2 * r].x, hq[2 * r + 1].x, hq[2 * r + 2].x, hq[2 * r + 3].x, hq[2 * r + 4].x);
            const unsigned o23 = pd_vert(hq[2 * r].y, hq[2 * r + 1].y, hq[2 * r + 2].y, hq[2 * r + 3].y, hq[2 * r + 4].y);
            const unsigned pack = __builtin_amdgcn_perm(o23, o01, 0x06040200u);
            uint8_t* dp = const_cast<uint8_t*>(d.p) + (size_t)(oy0 + r) * d.stride + ox0;
            if (dw_ok) *reinterpret_cast<uint32_t*>(dp) = pack;
            else for (int k = 0; k < cnt; k++) dp[k] = (uint8_t)(pack >> (8 * k));
        }
        return;
    }
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
#define RW_ROWS 1  // ROI rows per thread (measured at 256 streams: 4 rows 146 VGPRs / 3 waves per SIMD 630 us; 2 rows 75 VGPRs / 6 waves 499 us; 1 row 478 us)
#define RW_PX 8    // consecutive ROI pixels per thread and row (two packed dword stores)

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

// RNE(v) of a float32 |v| < 2^22 as ONE add: v + 1.5 * 2^23 has ulp 1, so the add rounds v to the nearest integer (ties to even, like cvRound),
// and the result's bit pattern is RW_MAGIC_I + round(v).  RW_MAGIC_I has its low 5 bits clear: (bits & 31) is already the 5-bit fraction.
#define RW_MAGIC 12582912.f
#define RW_MAGIC_I 0x4B400000
#define RW_RANGE 4000000.f

// the blends of one row of RW_PX pixels on the run path: T / B hold the source bytes of the top / bottom row from column c0 = sx0 - 1 on (byte j =
// column c0 + j).  Pixel k samples columns c0 + k + dk, dk in {0, 1, 2} (a zoom drifts the source column against k), i.e. two of the four bytes of the
// window starting at byte k: t = (32-ax) s00 + ax s01 is ONE v_dot4_u32_u8 of that window with the byte weights (32 - ax | ax << 8) shifted to byte dk --
// no byte selection, no widening.  Then (32-ay) t + ay b + 2^9 with the y weights pre-shifted by 6 bits, so that the result byte is byte 2.
__device__ __forceinline__ void rw_row_blend(const unsigned (&T)[4], const unsigned (&B)[4], const unsigned (&wxb)[RW_PX], const unsigned (&wlo)[RW_PX],
                                             const unsigned (&whi)[RW_PX], uint32_t& out0, uint32_t& out1)
{
    unsigned r[RW_PX];
#pragma unroll
    for (int k = 0; k < RW_PX; k++) {
        const int d = k >> 2, sh = k & 3;
        const unsigned tw = sh ? __builtin_amdgcn_alignbyte(T[d + 1], T[d], sh) : T[d];
        const unsigned bw = sh ? __builtin_amdgcn_alignbyte(B[d + 1], B[d], sh) : B[d];
        const unsigned t = __builtin_amdgcn_udot4(tw, wxb[k], 0u, false), b = __builtin_amdgcn_udot4(bw, wxb[k], 0u, false);
        r[k] = __umul24(b, whi[k]) + (__umul24(t, wlo[k]) + (1u << 15));  // ((32-ay) t + ay b + 2^9) << 6: the 8-bit result sits in bits 16..23
    }
    const unsigned p01 = __builtin_amdgcn_perm(r[1], r[0], 0x0c0c0602u), p23 = __builtin_amdgcn_perm(r[3], r[2], 0x06020c0cu);
    const unsigned p45 = __builtin_amdgcn_perm(r[5], r[4], 0x0c0c0602u), p67 = __builtin_amdgcn_perm(r[7], r[6], 0x06020c0cu);
    out0 = p01 | p23;
    out1 = p45 | p67;
}

typedef uint32_t rw_u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) rw_store8 { rw_u32x2 v; };
struct rw_load16 { unsigned a, b, c, d; };

// one thread: RW_PX pixels of RW_ROWS rows starting at (x8, ry0) of the ROI
__device__ __forceinline__ void rw_thread(const WarpJob& J, const ImgDesc& s, int rw, int rh, int x8, int ry0)
{
    if (ry0 >= rh) return;
    const int cnt = min(RW_PX, rw - x8);
    if (J.mode == 0) {
        for (int r = 0; r < RW_ROWS && ry0 + r < rh; r++) {
            const int sy = J.y0 + ry0 + r + J.dy;
            const bool yin = sy >= 0 && sy < s.h;
            for (int g = 0; g < cnt; g += 4) {
                const int c4 = min(4, cnt - g);
                uint32_t pack = 0;
                for (int k = 0; k < c4; k++) {
                    int sx = J.x0 + x8 + g + k + J.dx;
                    uint32_t v = (yin && sx >= 0 && sx < s.w) ? s.p[(size_t)sy * s.stride + sx] : 0u;
                    pack |= v << (8 * k);
                }
                roi_store4(J.dst + (size_t)(ry0 + r) * J.dst_stride, x8 + g, c4, pack);
            }
        }
        return;
    }
    // affine remap (numpy: x*T00 + y*T10 + T20 in float32, one rounding per operation, no fma; then cvRound(32 m)).  The map is evaluated 32 x
    // scaled: a power-of-two factor commutes with every float32 rounding, so fl(fl(32 x T00 + 32 y T10) + 32 T20) == 32 m bit for bit, and the
    // rounding to an integer is the single add of RW_MAGIC (see above) -- 3 adds per coordinate instead of 2 adds + mul + rndne + cvt.
    //
    // Near-identity maps (the tracker's case: translation + zoom + a little rotation): the RW_PX pixels of a row sample ONE source row pair, at
    // columns that run with k up to a drift of one column either way -- so the 2 x 11 source bytes come from two aligned 16-byte loads instead of 32
    // byte gathers ("run path").  Order of work: (A) the FIRST pixel of every row gives the load addresses (clamped into the frame, so the loads are
    // unconditional and leave at once), (B) the other coordinates, the run-path test and the weights are computed while the loads fly, (C) the
    // blends, (D) the stores -- or, for a thread whose rows do not qualify (ROI edge, strong rotation / zoom, map outside the frame), the general
    // per-pixel path.  No branch before (D): the straight-line part is what every interior thread runs.
    const float T0 = __fmul_rn(J.T[0], 32.f), T1 = __fmul_rn(J.T[1], 32.f), T2 = __fmul_rn(J.T[2], 32.f), T3 = __fmul_rn(J.T[3], 32.f),
                T4 = __fmul_rn(J.T[4], 32.f), T5 = __fmul_rn(J.T[5], 32.f);
    const float xbase = (float)(J.x0 + x8);
    float xa[RW_PX], xb[RW_PX];
    xa[0] = __fmul_rn(xbase, T0); xb[0] = __fmul_rn(xbase, T1);
    int fb[RW_ROWS][RW_PX], fyb[RW_ROWS][RW_PX];  // RW_MAGIC_I + fx, RW_MAGIC_I + fy
    int cb[RW_ROWS];                               // RW_MAGIC_I + 32 c0: fx[k] - 32 (c0 + k) = fb[k] - cb - 32 k
    float yx[RW_ROWS], yy[RW_ROWS];
    rw_load16 tq[RW_ROWS], bq[RW_ROWS];
    unsigned sh0[RW_ROWS], sh1[RW_ROWS];
    // (a thread at the right edge of the ROI owns fewer than 8 pixels: it computes all 8 -- the source window is clamped into the frame anyway -- and stores
    // cnt of them.  Round 3 sent it down the per-pixel path, which every fourth wavefront of a 1600-px ROI row then executed for ONE lane: 369 instead of
    // 251 instructions per working wavefront on average)
    unsigned bad = (ry0 + RW_ROWS > rh || s.w < 20 || s.h < 2) ? ~0u : 0u;
    if (((reinterpret_cast<uintptr_t>(J.dst) | (uintptr_t)J.dst_stride) & 3) != 0) bad = ~0u;  // packed dword stores need dword rows (x8 is a multiple of 8)
    const unsigned bsh = (unsigned)(reinterpret_cast<uintptr_t>(s.p) & 3);
    const uint8_t* bp = s.p - bsh;  // dword aligned, wave uniform
    // (A)
#pragma unroll
    for (int r = 0; r < RW_ROWS; r++) {
        const float y = (float)(J.y0 + ry0 + r);
        yx[r] = __fmul_rn(y, T2); yy[r] = __fmul_rn(y, T3);
        const float mx = __fadd_rn(__fadd_rn(xa[0], yx[r]), T4), my = __fadd_rn(__fadd_rn(xb[0], yy[r]), T5);
        // the magic add needs |32 m| < 2^22: checked on the first and (below) the last pixel of the row -- every operation of the map is monotone in x
        bad |= (fabsf(mx) < RW_RANGE && fabsf(my) < RW_RANGE) ? 0u : ~0u;
        fb[r][0] = __float_as_int(__fadd_rn(mx, RW_MAGIC));
        fyb[r][0] = __float_as_int(__fadd_rn(my, RW_MAGIC));
        const int c0 = ((fb[r][0] - RW_MAGIC_I) >> 5) - 1, sy0 = (fyb[r][0] - RW_MAGIC_I) >> 5;
        // source window inside the frame: the aligned 16-byte loads cover columns c0 - 3 .. c0 + 15
        const int cx = max(min(max(c0, 3), s.w - 16), 0), cy = max(min(max(sy0, 0), s.h - 2), 0);  // (frames narrower than 20 px: the thread is `bad`, the address stays at the row start)
        bad |= (cx != c0 || cy != sy0) ? ~0u : 0u;
        cb[r] = (fb[r][0] & ~31) - 32;
        const unsigned o0 = (unsigned)(__mul24(cy, s.stride) + cx) + bsh, o1 = o0 + (unsigned)s.stride;
        sh0[r] = o0 & 3u; sh1[r] = o1 & 3u;
        pd_gptr p0 = (pd_gptr)(bp + (o0 & ~3u)), p1 = (pd_gptr)(bp + (o1 & ~3u));
        tq[r].a = p0[0]; tq[r].b = p0[1]; tq[r].c = p0[2]; tq[r].d = p0[3];
        bq[r].a = p1[0]; bq[r].b = p1[1]; bq[r].c = p1[2]; bq[r].d = p1[3];
    }
    __builtin_amdgcn_sched_barrier(0);  // all 8 loads leave here: the scheduler otherwise sinks each row's pair next to its blend (4 serial round trips)
    // (B)
#pragma unroll
    for (int k = 1; k < RW_PX; k++) {
        const float x = xbase + (float)k;  // exact (integers below 2^24)
        xa[k] = __fmul_rn(x, T0); xb[k] = __fmul_rn(x, T1);
    }
    unsigned wxb[RW_ROWS][RW_PX];
#pragma unroll
    for (int r = 0; r < RW_ROWS; r++) {
        float mx7 = 0.f, my7 = 0.f;
#pragma unroll
        for (int k = 1; k < RW_PX; k++) {
            const float mx = __fadd_rn(__fadd_rn(xa[k], yx[r]), T4), my = __fadd_rn(__fadd_rn(xb[k], yy[r]), T5);
            fb[r][k] = __float_as_int(__fadd_rn(mx, RW_MAGIC));
            fyb[r][k] = __float_as_int(__fadd_rn(my, RW_MAGIC));
            if (k == RW_PX - 1) { mx7 = mx; my7 = my; }
        }
        // e_k = fx[k] - 32 (c0 + k) must lie in [0, 96): source column c0 + k + dk with dk = e_k >> 5 in {0, 1, 2} (e_0 is in [32, 64) by construction);
        // (fy[k] >> 5) == sy0 for every k  <=>  for k = 7 (monotone)
        unsigned over = (unsigned)(fyb[r][RW_PX - 1] ^ fyb[r][0]) >> 5;
#pragma unroll
        for (int k = 0; k < RW_PX; k++) {
            const unsigned e = (unsigned)(fb[r][k] - cb[r] - 32 * k);
            over |= (e >= 96u) ? 1u : 0u;
            // byte weights (32 - ax | ax << 8) = 255 ax + 32, moved to byte dk of the window that starts at byte k
            wxb[r][k] = (__umul24((unsigned)(fb[r][k] & 31), 255u) + 32u) << ((e >> 2) & 0x18u);
        }
        bad |= (over == 0u && fabsf(mx7) < RW_RANGE && fabsf(my7) < RW_RANGE) ? 0u : ~0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    // (C)
    uint32_t res[RW_ROWS][2];
#pragma unroll
    for (int r = 0; r < RW_ROWS; r++) {
        unsigned T[4], B[4], wlo[RW_PX], whi[RW_PX];
        T[0] = __builtin_amdgcn_alignbyte(tq[r].b, tq[r].a, sh0[r]); T[1] = __builtin_amdgcn_alignbyte(tq[r].c, tq[r].b, sh0[r]);
        T[2] = __builtin_amdgcn_alignbyte(tq[r].d, tq[r].c, sh0[r]); T[3] = tq[r].d >> (8 * sh0[r]);
        B[0] = __builtin_amdgcn_alignbyte(bq[r].b, bq[r].a, sh1[r]); B[1] = __builtin_amdgcn_alignbyte(bq[r].c, bq[r].b, sh1[r]);
        B[2] = __builtin_amdgcn_alignbyte(bq[r].d, bq[r].c, sh1[r]); B[3] = bq[r].d >> (8 * sh1[r]);
#pragma unroll
        for (int k = 0; k < RW_PX; k++) {
            const unsigned ay64 = (unsigned)(fyb[r][k] & 31) << 6;
            wlo[k] = 2048u - ay64; whi[k] = ay64;
        }
        rw_row_blend(T, B, wxb[r], wlo, whi, res[r][0], res[r][1]);
    }
    // (D)
    if (bad == 0u) {
        if (cnt == RW_PX) {
#pragma unroll
            for (int r = 0; r < RW_ROWS; r++) {
                rw_store8 o;
                o.v = rw_u32x2{res[r][0], res[r][1]};
                *reinterpret_cast<rw_store8*>(J.dst + (size_t)(ry0 + r) * J.dst_stride + x8) = o;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RW_ROWS; r++) {
                uint8_t* drow = J.dst + (size_t)(ry0 + r) * J.dst_stride;
                roi_store4(drow, x8, min(4, cnt), res[r][0]);
                if (cnt > 4) roi_store4(drow, x8 + 4, cnt - 4, res[r][1]);
            }
        }
        return;
    }
    // general path (ROI edge columns / rows, strong rotations, maps far outside the frame): per-pixel gathers, coordinates exactly as numpy + cvRound
    for (int r = 0; r < RW_ROWS && ry0 + r < rh; r++) {
        const float y = (float)(J.y0 + ry0 + r);
        const float yx = __fmul_rn(y, J.T[2]), yy = __fmul_rn(y, J.T[3]);
        for (int g = 0; g < cnt; g += 4) {
            const int c4 = min(4, cnt - g);
            uint32_t pack = 0;
            for (int k = 0; k < c4; k++) {
                const float x = (float)(J.x0 + x8 + g + k);
                const float mx = __fadd_rn(__fadd_rn(__fmul_rn(x, J.T[0]), yx), J.T[4]);
                const float my = __fadd_rn(__fadd_rn(__fmul_rn(x, J.T[1]), yy), J.T[5]);
                const int fx = vh_round(__fmul_rn(mx, 32.f)), fy = vh_round(__fmul_rn(my, 32.f));
                const int sx = fx >> 5, sy = fy >> 5;
                const bool x0in = sx >= 0 && sx < s.w, x1in = sx + 1 >= 0 && sx + 1 < s.w;
                const bool y0in = sy >= 0 && sy < s.h, y1in = sy + 1 >= 0 && sy + 1 < s.h;
                const uint8_t* r0 = s.p + (ptrdiff_t)sy * s.stride + sx;
                const int s00 = (y0in && x0in) ? r0[0] : 0, s01 = (y0in && x1in) ? r0[1] : 0;
                const int s10 = (y1in && x0in) ? r0[s.stride] : 0, s11 = (y1in && x1in) ? r0[s.stride + 1] : 0;
                pack |= remap_blend(s00, s01, s10, s11, fx & 31, fy & 31) << (8 * k);
            }
            roi_store4(J.dst + (size_t)(ry0 + r) * J.dst_stride, x8 + g, c4, pack);
        }
    }
}

#define RW_LOOP 1  // row groups a thread walks one after the other (4: 533 us against 483 -- the loop serialises the load round trips; wave start-up is not the cost)
__global__ __launch_bounds__(256) void k_roi_warp(const void* job_tab, size_t tab_stride, int xcd_bands)
{
    // (an XCD-contiguous block remap measured 478 -> 628 us here: the dispatcher's round robin spreads every ROI over all channels)
    const WarpJob J = *reinterpret_cast<const WarpJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blockIdx.z * tab_stride);
    if (J.mode < 0) return;
    const int rw = J.x1 - J.x0, rh = J.y1 - J.y0;
    // EXPERIMENT SWITCH (VH_RW_XCD=1, off by default; round 6): XCD-aware tile order.  A workgroup's 128-px tile row is one 128-byte line only when the
    // source column is line aligned; an ROI starts at any column, so nearly every tile row straddles two lines and shares each with its left / right
    // neighbour -- and the dispatcher deals consecutive workgroups (x fastest) round-robin to the 8 XCDs, so the two workgroups that share a line sit on
    // different L2s and both fetch it from the fabric.  With xcd_bands the linear workgroup id L (gridDim.y padded to a multiple of 8: L mod 8 is the XCD)
    // is re-indexed so that XCD q walks tile row band 8 m + q from left to right.  Measured: the fabric reads fall as predicted,
    // the kernel gets SLOWER -- it is latency / issue bound, not traffic bound -- so the natural order stays the default.
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (xcd_bands) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, k = L >> 3, band = k / gridDim.x;
        bx = k - band * gridDim.x;
        by = band * 8u + (L & 7u);
    }
    const int x8 = (int)(bx * blockDim.x + threadIdx.x) * RW_PX;
    if (x8 >= rw) return;
    const ImgDesc s = J.src;
    // consecutive rows of a block stay adjacent (threadIdx.y), the RW_LOOP passes of a block are blockDim.y * RW_ROWS rows apart
    const int yb = (int)by * (int)blockDim.y * RW_ROWS * RW_LOOP + (int)threadIdx.y * RW_ROWS;
    if (yb >= rh) return;
#pragma unroll 1
    for (int l = 0; l < RW_LOOP; l++) rw_thread(J, s, rw, rh, x8, yb + l * (int)blockDim.y * RW_ROWS);
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

// Fused ingest: ONE pass over the BGR frame writes the gray frame (cvtColor, vidExample.py:91) AND the quarter-scale image
// KLTmain starts from (cv2.resize(im, (0,0), fx=.25, fy=.25, INTER_NEAREST), KLT.py:111-113): small[y][x] = gray[4 y][4 x] (dsize = round(src / 4), so
// 4 x <= w - 1 always and the min() of the resize never binds).  Same thread layout as k_bgr2gray; the threads of rows 4 y additionally store one
// quarter-scale pixel each.  `jobs`: one descriptor per frame (blockIdx.z).
__global__ __launch_bounds__(256) void k_ingest_bgr(const IngestJob* jobs)
{
    const IngestJob J = jobs[blockIdx.z];
    if (J.bgr == nullptr) return;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= J.h || x4 >= J.w) return;
    const uint8_t* s = J.bgr + (size_t)y * J.bgr_stride + 3 * (size_t)x4;
    uint8_t* d = J.gray + (size_t)y * J.gray_stride + x4;
    const int cnt = min(4, J.w - x4);
    uint32_t pack = 0;
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(s) & 3) == 0)) {  // 12 aligned bytes: three dword loads
        const uint32_t a = reinterpret_cast<const uint32_t*>(s)[0], b = reinterpret_cast<const uint32_t*>(s)[1], c = reinterpret_cast<const uint32_t*>(s)[2];
        const uint32_t px[4][3] = {{a & 255u, (a >> 8) & 255u, (a >> 16) & 255u}, {a >> 24, b & 255u, (b >> 8) & 255u},
                                   {(b >> 16) & 255u, b >> 24, c & 255u}, {(c >> 8) & 255u, (c >> 16) & 255u, c >> 24}};
#pragma unroll
        for (int k = 0; k < 4; k++) pack |= ((px[k][0] * 3735u + px[k][1] * 19235u + px[k][2] * 9798u + (1u << 14)) >> 15) << (8 * k);
    } else {
        for (int k = 0; k < cnt; k++) pack |= ((s[3 * k] * 3735u + s[3 * k + 1] * 19235u + s[3 * k + 2] * 9798u + (1u << 14)) >> 15) << (8 * k);
    }
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) *reinterpret_cast<uint32_t*>(d) = pack;
    else for (int k = 0; k < cnt; k++) d[k] = (uint8_t)(pack >> (8 * k));
    if (J.small != nullptr && (y & 3) == 0 && (x4 >> 2) < J.dw && (y >> 2) < J.dh) J.small[(size_t)(y >> 2) * J.small_stride + (x4 >> 2)] = (uint8_t)pack;
}

void vh_launch_ingest_bgr(const IngestJob* jobs_dev, int count, int max_w, int max_h, hipStream_t s)
{
    dim3 blk(64, 4), grd((max_w + 255) / 256, (max_h + 3) / 4, count);
    hipLaunchKernelGGL(k_ingest_bgr, grd, blk, 0, s, jobs_dev);
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
    dim3 blk(64, 4), grd((max_dw + 255) / 256, (max_dh + 4 * RQ_ROWS - 1) / (4 * RQ_ROWS), batch * per_stream);
    hipLaunchKernelGGL(k_resize_quarter, grd, blk, 0, s, src_tab, dst_tab, tab_stride, per_stream);
}

// REFLECT_101 border ring of a freshly built small level (see VH_LV_PAD): one thread per dword of 4 ring pixels (the interior origin and the row
// pitch are dword aligned; a dword that straddles the image edge re-writes interior pixels with themselves).  Same build table as k_pyr_down;
// levels without a border (large levels, user buffers) and missing levels exit at once.
__global__ __launch_bounds__(256) void k_pyr_pad(const void* pb_tab, size_t ws_stride, int lvl)
{
    const PyrBuild& pb = reinterpret_cast<const PyrBuild*>(reinterpret_cast<const char*>(pb_tab) + (size_t)(blockIdx.z >> 1) * ws_stride)[blockIdx.z & 1];
    if (!pb.enable || pb.pyr == nullptr) return;
    const pd_gdesc Pg = (pd_gdesc)pb.pyr;  // (global, not generic: scalar loads, see k_pyr_down)
    if (lvl + 1 >= Pg->nlevels) return;
    const ImgDesc d = pd_level(Pg, lvl + 1);
    const int B = d.pad;
    if (B <= 0) return;
    const int wq = (d.w + 2 * B + 3) >> 2;        // dwords of a full-width ring row (columns -B .. w+B-1, rounded up: the pitch has 4 spare bytes)
    const int rq = ((d.w & 3) + B + 3) >> 2;       // dwords of the right band of an image row, starting at column w & ~3
    const int band = wq * 2 * B, side = (B / 4) + rq;
    // A FIXED small grid per image (gridDim.x workgroups, each walks the ring with a stride of gridDim.x x 256 dwords) and a loop, not one thread per ring dword of the largest bordered level: most launches of a step find levels
    // without a ring (the ROI pyramids' levels are large) and used to dispatch ~12 000 workgroups that left at once -- 10 us per launch, four per step
    for (int t = blockIdx.x * 256 + threadIdx.x; t < band + d.h * side; t += (int)gridDim.x * 256) {
        int x, y;
        if (t < band) {  // the 2 B full-width rows above and below
            const int q = t / wq;
            x = 4 * (t - q * wq) - B;
            y = q < B ? q - B : d.h + q - B;
        } else {         // left (B / 4 dwords) and right (rq dwords) of the image rows
            const int u = t - band;
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
}

static std::atomic<int> g_pyr_rows{0};  // PROCESS-WIDE test hook (include/velocity_hip.h): 2 / 4 / 8 output rows per thread whatever the launch size (0: by size)
void vh_pyr_force_rows(int rb) { g_pyr_rows.store(rb, std::memory_order_relaxed); }

void vh_launch_pyr_down_ws(const void* pb_tab, size_t ws_stride, int batch, int lvl, int max_w0, int max_h0, hipStream_t s)
{
    // dims of level lvl+1 when level 0 is max_w0 x max_h0
    int w = max_w0, h = max_h0;
    for (int l = 0; l <= lvl; l++) { w = (w + 1) / 2; h = (h + 1) / 2; }
    // 64 x 4 threads: a wavefront is ONE row of 256 output pixels.  Wavefronts of 16 x 4 or 32 x 2 threads (fewer idle lanes in the last column of
    // workgroups: 818 output pixels are 3.2 wavefronts of 256) measured SLOWER -- 451 / 454 against 405 us for pyrDown + rings at 256 streams: a
    // thread's 19 source rows are shared with no other row of its wavefront, and narrower row segments coalesce worse
    const dim3 blk(64, 4);
    // 8 output rows per thread (19 source rows in flight, 136 VGPRs) once a launch is far beyond the chip (19 rows read per 16 produced instead of
    // 11 per 8: 271 -> 250 us for the 766 x 451 level of 256 streams; no gain below), 4 rows from ~1 Mpx, 2 for single-stream latency
    const long long px = (long long)w * h * batch;
    const int forced_rows = g_pyr_rows.load(std::memory_order_relaxed);
    const int rb = forced_rows ? forced_rows : (px >= (1ll << 25) ? 8 : px >= (1ll << 20) ? 4 : 2);
    if (rb == 8) {
        dim3 grd((w + 4 * blk.x - 1) / (4 * blk.x), (h + 8 * blk.y - 1) / (8 * blk.y), batch * 2);
        hipLaunchKernelGGL(k_pyr_down<8>, grd, blk, 0, s, pb_tab, ws_stride, lvl);
    } else if (rb == 4) {
        dim3 grd((w + 4 * blk.x - 1) / (4 * blk.x), (h + 4 * blk.y - 1) / (4 * blk.y), batch * 2);
        hipLaunchKernelGGL(k_pyr_down<4>, grd, blk, 0, s, pb_tab, ws_stride, lvl);
    } else {
        dim3 grd((w + 4 * blk.x - 1) / (4 * blk.x), (h + 2 * blk.y - 1) / (2 * blk.y), batch * 2);
        hipLaunchKernelGGL(k_pyr_down<2>, grd, blk, 0, s, pb_tab, ws_stride, lvl);
    }
    // border ring of the new level when it is a small one (decided per image on the device): a few workgroups per image, each loops over its share --
    // 4 when the launch has many images (327 against 338 us for pyrDown + rings at 256 streams), 32 for a few streams (a ring of <= ~7000 dwords in one pass:
    // with 4 a single-stream step took 6 us longer)
    hipLaunchKernelGGL(k_pyr_pad, dim3(batch >= 16 ? 4 : 32, 1, batch * 2), dim3(256), 0, s, pb_tab, ws_stride, lvl);
}

void vh_launch_roi_warp(const void* job_tab, size_t tab_stride, int batch, int max_w, int max_h, hipStream_t s)
{
    // 16 x 16 threads: a wavefront covers 4 consecutive ROI rows of 128 pixels, so the bottom source row of one thread row is the top source row of
    // the next INSIDE the wavefront (one L1 fetch instead of two) and the last column of workgroups idles 28 of 1636 pixels instead of 412.  Measured at
    // 256 streams (A/B on one box): 64 x 4 threads (one 512-pixel row per wavefront) 370 us, 32 x 8 375, 16 x 16 301, 16 x 8 292-302, 8 x 32 305
    const dim3 blk(16, 16);
    static const int xcd_bands = getenv("VH_RW_XCD") ? atoi(getenv("VH_RW_XCD")) : 0;  // (environment: experiments only; see the kernel)
    const unsigned rows = (max_h + blk.y * RW_ROWS * RW_LOOP - 1) / (blk.y * RW_ROWS * RW_LOOP);
    const dim3 grd((max_w + blk.x * RW_PX - 1) / (blk.x * RW_PX), xcd_bands ? (rows + 7u) & ~7u : rows, batch);  // (padded rows: workgroups beyond the ROI leave at once)
    hipLaunchKernelGGL(k_roi_warp, grd, blk, 0, s, job_tab, tab_stride, xcd_bands);
}
