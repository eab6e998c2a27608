// Image-stage kernels of the KLT path: quarter-scale nearest decimation (K1), pyrDown (K2), shifted crop (K8) and
// affine remap (K9).  All are HBM-bound byte kernels; descriptors are read from device memory because ROI sizes are
// data dependent (bounding box of the tracks) and the whole frame pipeline runs without a host round trip.
#include "vh_kernels.hpp"

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
    for (int k = 0; k < cnt; k++) pack |= (uint32_t)srow[min(4 * (x4 + k), s.w - 1)] << (8 * k);
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(drow + x4) & 3) == 0)) {
        *reinterpret_cast<uint32_t*>(drow + x4) = pack;
    } else {
        for (int k = 0; k < cnt; k++) drow[x4 + k] = (uint8_t)(pack >> (8 * k));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pyrDown: dst = ((w+1)/2, (h+1)/2), separable [1 4 6 4 1], REFLECT_101, (sum + 128) >> 8   
// Block = 256 threads, output tile 64 x 16.  Source tile (35 rows x 131 cols) is staged in LDS once, the
// horizontal pass writes 35 x 64 int16 partial sums to LDS, the vertical pass produces the tile.
// Build tables: stream b = blockIdx.z / 2 owns two PyrBuild entries (previous / current image); level `lvl` of the
// pyramid is read, level lvl+1 is written; disabled entries and pyramids with fewer levels exit immediately.
// ---------------------------------------------------------------------------------------------------------------
#define PD_TW 64
#define PD_TH 16
#define PD_SW (2 * PD_TW + 3)
#define PD_SH (2 * PD_TH + 3)
#define PD_SP 136  // padded LDS row pitch (bytes)

__global__ __launch_bounds__(256) void k_pyr_down(const void* pb_tab, size_t ws_stride, int lvl)
{
    const PyrBuild& pb = reinterpret_cast<const PyrBuild*>(reinterpret_cast<const char*>(pb_tab) + (size_t)(blockIdx.z >> 1) * ws_stride)[blockIdx.z & 1];
    if (!pb.enable || pb.pyr == nullptr) return;
    const PyrDesc& P = *pb.pyr;
    if (lvl + 1 >= P.nlevels) return;
    const ImgDesc s = P.lv[lvl];
    const ImgDesc d = P.lv[lvl + 1];
    const int ox0 = blockIdx.x * PD_TW, oy0 = blockIdx.y * PD_TH;
    if (ox0 >= d.w || oy0 >= d.h) return;

    __shared__ uint8_t tile[PD_SH * PD_SP];
    __shared__ uint16_t hsum[PD_SH * PD_TW];
    const int tid = threadIdx.x;
    const int sx0 = 2 * ox0 - 2, sy0 = 2 * oy0 - 2;
    const bool interior = sx0 >= 0 && sy0 >= 0 && sx0 + PD_SW <= s.w && sy0 + PD_SH <= s.h;
    if (interior) {
        for (int i = tid; i < PD_SH * PD_SW; i += 256) {
            int r = i / PD_SW, c = i - r * PD_SW;
            tile[r * PD_SP + c] = s.p[(size_t)(sy0 + r) * s.stride + sx0 + c];
        }
    } else {
        for (int i = tid; i < PD_SH * PD_SW; i += 256) {
            int r = i / PD_SW, c = i - r * PD_SW;
            int yy = vh_reflect101(sy0 + r, s.h), xx = vh_reflect101(sx0 + c, s.w);
            tile[r * PD_SP + c] = s.p[(size_t)yy * s.stride + xx];
        }
    }
    __syncthreads();
    for (int i = tid; i < PD_SH * PD_TW; i += 256) {
        int r = i >> 6, c = i & 63;
        const uint8_t* t = &tile[r * PD_SP + 2 * c];
        hsum[i] = (uint16_t)(t[0] + 4 * t[1] + 6 * t[2] + 4 * t[3] + t[4]);
    }
    __syncthreads();
    for (int i = tid; i < PD_TH * PD_TW; i += 256) {
        int r = i >> 6, c = i & 63;
        int ox = ox0 + c, oy = oy0 + r;
        if (ox < d.w && oy < d.h) {
            const uint16_t* hcol = &hsum[(2 * r) * PD_TW + c];
            int acc = hcol[0] + 4 * hcol[PD_TW] + 6 * hcol[2 * PD_TW] + 4 * hcol[3 * PD_TW] + hcol[4 * PD_TW];
            const_cast<uint8_t*>(d.p)[(size_t)oy * d.stride + ox] = (uint8_t)((acc + 128) >> 8);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ROI warp stage of KLTregional (utils/KLT.py:65-73).  mode 0: integer-shifted crop, zero outside the frame
// (translateFlag branch, SURVEY App. B intent); mode 1: float32 affine map + remap(INTER_LINEAR) with 5-bit
// fixed-point coordinates and 15-bit weights, constant-0 border.  One thread = 4 consecutive ROI pixels.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_roi_warp(const void* job_tab, size_t tab_stride)
{
    const WarpJob J = *reinterpret_cast<const WarpJob*>(reinterpret_cast<const char*>(job_tab) + (size_t)blockIdx.z * tab_stride);
    if (J.mode < 0) return;
    const int rw = J.x1 - J.x0, rh = J.y1 - J.y0;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int ry = blockIdx.y * blockDim.y + threadIdx.y;
    if (ry >= rh || x4 >= rw) return;
    const ImgDesc s = J.src;
    uint8_t* drow = J.dst + (size_t)ry * J.dst_stride;
    const int cnt = min(4, rw - x4);
    uint32_t pack = 0;
    if (J.mode == 0) {
        const int sy = J.y0 + ry + J.dy;
        const bool yin = sy >= 0 && sy < s.h;
        for (int k = 0; k < cnt; k++) {
            int sx = J.x0 + x4 + k + J.dx;
            uint32_t v = (yin && sx >= 0 && sx < s.w) ? s.p[(size_t)sy * s.stride + sx] : 0u;
            pack |= v << (8 * k);
        }
    } else {
        const float y = (float)(J.y0 + ry);
        for (int k = 0; k < cnt; k++) {
            const float x = (float)(J.x0 + x4 + k);
            // float32, one rounding per operation, no fma (numpy: x*T00 + y*T10 + T20)
            const float mx = __fadd_rn(__fadd_rn(__fmul_rn(x, J.T[0]), __fmul_rn(y, J.T[2])), J.T[4]);
            const float my = __fadd_rn(__fadd_rn(__fmul_rn(x, J.T[1]), __fmul_rn(y, J.T[3])), J.T[5]);
            const int fx = vh_round(__fmul_rn(mx, 32.f)), fy = vh_round(__fmul_rn(my, 32.f));
            const int sx = fx >> 5, sy = fy >> 5, ax = fx & 31, ay = fy & 31;
            const bool x0in = sx >= 0 && sx < s.w, x1in = sx + 1 >= 0 && sx + 1 < s.w;
            const bool y0in = sy >= 0 && sy < s.h, y1in = sy + 1 >= 0 && sy + 1 < s.h;
            const uint8_t* r0 = s.p + (ptrdiff_t)sy * s.stride + sx;
            const int s00 = (y0in && x0in) ? r0[0] : 0, s01 = (y0in && x1in) ? r0[1] : 0;
            const int s10 = (y1in && x0in) ? r0[s.stride] : 0, s11 = (y1in && x1in) ? r0[s.stride + 1] : 0;
            const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
            const uint32_t v = (uint32_t)((s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11 + (1 << 14)) >> 15);
            pack |= v << (8 * k);
        }
    }
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(drow + x4) & 3) == 0)) {
        *reinterpret_cast<uint32_t*>(drow + x4) = pack;
    } else {
        for (int k = 0; k < cnt; k++) drow[x4 + k] = (uint8_t)(pack >> (8 * k));
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
// host launchers
// ---------------------------------------------------------------------------------------------------------------
void vh_launch_resize_quarter(const void* src_tab, const void* dst_tab, size_t tab_stride, int per_stream, int batch, int max_dw, int max_dh,
                              hipStream_t s)
{
    dim3 blk(64, 4), grd((max_dw + 255) / 256, (max_dh + 3) / 4, batch * per_stream);
    hipLaunchKernelGGL(k_resize_quarter, grd, blk, 0, s, src_tab, dst_tab, tab_stride, per_stream);
}

void vh_launch_pyr_down_ws(const void* pb_tab, size_t ws_stride, int batch, int lvl, int max_w0, int max_h0, hipStream_t s)
{
    // dims of level lvl+1 when level 0 is max_w0 x max_h0
    int w = max_w0, h = max_h0;
    for (int l = 0; l <= lvl; l++) { w = (w + 1) / 2; h = (h + 1) / 2; }
    dim3 grd((w + PD_TW - 1) / PD_TW, (h + PD_TH - 1) / PD_TH, batch * 2);
    hipLaunchKernelGGL(k_pyr_down, grd, dim3(256), 0, s, pb_tab, ws_stride, lvl);
}

void vh_launch_roi_warp(const void* job_tab, size_t tab_stride, int batch, int max_w, int max_h, hipStream_t s)
{
    dim3 blk(64, 4), grd((max_w + 255) / 256, (max_h + 3) / 4, batch);
    hipLaunchKernelGGL(k_roi_warp, grd, blk, 0, s, job_tab, tab_stride);
}
