// Device-side glue of KLTmain between its stages (utils/KLT.py:121-133): job descriptors of the next stage written by ONE workgroup per stream.
// Shared by vh_api.hip (k_klt_setup, k_klt_glue1 / k_klt_glue2, k_regional_setup) and vh_ransac.hip: the fused RANSAC kernel -- also one workgroup per
// stream -- runs the glue that follows it as its own epilogue (round 6: two dependent launches fewer per frame step; a launch boundary costs a single
// stream ~4.5 us).
#pragma once
#include "vh_ws.hpp"

__device__ inline void fill_pyramid(PyrDesc& P, const uint8_t* lv0, int w, int h, int stride, uint8_t* const* lvbuf, int win, int max_level)
{
    // OpenCV truncation: keep level L, stop when level L+1 would be <= win in either dimension
    int n = 0;
    for (int level = 0; level <= max_level && level < VH_MAX_LEVELS; level++) {
        ImgDesc& d = P.lv[level];
        if (level == 0) { d.p = lv0; d.w = w; d.h = h; d.stride = stride; d.pad = 0; }
        else if (VH_LV_PADDED(w, h)) { d.stride = VH_LV_STRIDE(w); d.p = lvbuf[level] + (size_t)VH_LV_PAD * d.stride + VH_LV_PAD; d.w = w; d.h = h; d.pad = VH_LV_PAD; }
        else { d.p = lvbuf[level]; d.w = w; d.h = h; d.stride = (w + 3) & ~3; d.pad = 0; }  // dword rows: k_pyr_down stores packed dwords (byte stores on a 766-pixel pitch cost it 2x)
        n = level + 1;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win || h <= win) break;
    }
    P.nlevels = n;
}

__device__ inline void clamp_criteria(const vh_lk_params& lk, int& max_count, double& eps2)
{
    max_count = min(max(lk.max_count, 0), 100);
    double e = fmin(fmax(lk.eps, 0.0), 10.0);
    eps2 = e * e;
}

__device__ inline void fill_lk_common(LKJob& J, const vh_lk_params& lk, const float* p_in, const int* n_ptr, int n)
{
    J.p_in = p_in; J.n_ptr = n_ptr; J.n = n; J.stats = nullptr; J.order = nullptr;
    J.win = lk.win; J.max_level = lk.max_level;
    clamp_criteria(lk, J.max_count, J.eps2);
    J.err_out = nullptr; J.fbe_out = nullptr; J.praw_out = nullptr;
}


// ---- stage 1 -> 2: mean translation, ROI, shifted crop, job B (KLT.py:121-124, 55-68) ---------------------------
template <int NW>
__device__ __forceinline__ float block_min_f(float v, float* sh /* [NW] */, bool is_max)
{
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, t) : fminf(v, t);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
    for (int k = 1; k < NW; k++) r = is_max ? fmaxf(r, sh[k]) : fminf(r, sh[k]);
    return r;
}

// NT threads of ONE workgroup per stream: k_klt_glue1 (256) or the tail of k_ransac_fused<1> (512)
template <int NT>
__device__ __forceinline__ void klt_glue1_body(StreamWS& ws)
{
    constexpr int NW = NT / 64;
    const KltIO& io = ws.io;
    const StreamBufs& B = ws.bufs;
    const int n = ws.n, tid = threadIdx.x;
    __shared__ long long sh_i[3 * NW];
    __shared__ float sh_f[NW];
    long long s[3] = {0, 0, 0};
    float mnx = 3.0e38f, mny = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f;
    for (int i = tid; i < n; i += NT) {
        const float x0 = io.p0[2 * i], y0 = io.p0[2 * i + 1];
        mnx = fminf(mnx, x0); mxx = fmaxf(mxx, x0); mny = fminf(mny, y0); mxy = fmaxf(mxy, y0);
        if (B.v_small[i]) {
            const float dx = __fsub_rn(B.p_small[2 * i], x0), dy = __fsub_rn(B.p_small[2 * i + 1], y0);
            s[0] += vh_fixq((double)dx, 32); s[1] += vh_fixq((double)dy, 32); s[2] += 1;
        }
    }
    // block reductions
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int k = 0; k < 3; k++) s[k] = vh_wave_sum_i64(s[k]);
        if (lane == 0) for (int k = 0; k < 3; k++) sh_i[k * NW + wave] = s[k];
        __syncthreads();
        for (int k = 0; k < 3; k++) {
            long long t = 0;
            for (int w = 0; w < NW; w++) t += sh_i[k * NW + w];
            s[k] = t;
        }
    }
    mnx = block_min_f<NW>(mnx, sh_f, false); mny = block_min_f<NW>(mny, sh_f, false);
    mxx = block_min_f<NW>(mxx, sh_f, true);  mxy = block_min_f<NW>(mxy, sh_f, true);
    __shared__ int s_copy, s_box[6];
    if (tid == 0) s_copy = 0;
    __syncthreads();
    if (tid == 0) {

    const double cnt = (double)s[2];
    const double tx = s[2] ? __ddiv_rn(ldexp((double)s[0], -32), cnt) : 0.0;
    const double ty = s[2] ? __ddiv_rn(ldexp((double)s[1], -32), cnt) : 0.0;
    ws.t_trans[0] = tx; ws.t_trans[1] = ty;
    // boundingRect(p0, im.shape, border=(50,50))  (images.py:9-19, KLT.py:60)
    int x0 = vh_floor(mnx), y0 = vh_floor(mny);
    const int bw = vh_floor(mxx) - x0 + 1, bh = vh_floor(mxy) - y0 + 1;
    int x1 = x0 + bw + 50, y1 = y0 + bh + 50;
    x0 -= 50; y0 -= 50;
    x0 = max(x0, 1); y0 = max(y0, 1); x1 = min(x1, io.w); y1 = min(y1, io.h);
    if (n == 0) { x0 = 1; y0 = 1; x1 = 1; y1 = 1; }
    ws.roi[0] = x0; ws.roi[1] = x1; ws.roi[2] = y0; ws.roi[3] = y1;
    const int rw = max(x1 - x0, 0), rh = max(y1 - y0, 0);
    const int dx = (int)(float)tx, dy = (int)(float)ty;  // T.astype(float32)[2].__int__()  (KLT.py:58,66-67)
    ws.dxy[0] = dx; ws.dxy[1] = dy;
    // shifted crop: a view when it stays inside the frame, a zero-padded copy otherwise
    WarpJob& W = ws.warp;
    W.src = ImgDesc{io.im, io.w, io.h, io.stride, 0};
    W.dst = B.warp; W.dst_stride = (rw + 3) & ~3;  // dword rows: the warp kernel stores packed dwords
    W.x0 = x0; W.x1 = x1; W.y0 = y0; W.y1 = y1; W.dx = dx; W.dy = dy;
    const bool inside = x0 + dx >= 0 && x1 + dx <= io.w && y0 + dy >= 0 && y1 + dy <= io.h;
    // the copy (rare: the shifted crop leaves the frame) is done by this workgroup below, not by a k_roi_warp launch that is a no-op on almost
    // every frame (5 us of dependent launch latency for one stream, 30 us at 128 streams)
    W.mode = -1;
    s_copy = (inside || n == 0) ? 0 : 1;
    s_box[0] = x0; s_box[1] = x1; s_box[2] = y0; s_box[3] = y1; s_box[4] = dx; s_box[5] = dy;
    LKJob& J = ws.lk;
    const vh_lk_params& lk = io.coarse;
    fill_pyramid(J.I, io.im0 + (size_t)y0 * io.stride0 + x0, rw, rh, io.stride0, B.roi_lv[0], lk.win, lk.max_level);
    if (inside) fill_pyramid(J.J, io.im + (ptrdiff_t)(y0 + dy) * io.stride + (x0 + dx), rw, rh, io.stride, B.roi_lv[1], lk.win, lk.max_level);
    else fill_pyramid(J.J, B.warp, rw, rh, (rw + 3) & ~3, B.roi_lv[1], lk.win, lk.max_level);
    ws.pb[0] = PyrBuild{&J.I, 1, 0};
    ws.pb[1] = PyrBuild{&J.J, 1, 0};
    fill_lk_common(J, lk, io.p0, nullptr, n);
    J.order = ws.order;
    J.p_out = B.p_coarse; J.v_out = B.v_coarse;
    J.fbt = io.fbt_coarse;
    J.in_scale = 1.f; J.in_off[0] = (float)x0; J.in_off[1] = (float)y0;
    J.out_mode = VH_OUT_TRANSLATE; J.out_off[0] = (float)dx; J.out_off[1] = (float)dy;
    J.stats = &ws.lk_stats[1][0][0];
    // RANSAC 2: affine from the survivors, only when more than 10 of them (KLT.py:126-127)
    RansacJob& R = ws.ransac;
    R.to = B.p_coarse; R.valid = B.v_coarse; R.min_valid = 10; R.gate_valid = 0;
    }
    __syncthreads();
    if (s_copy) {  // integer-shifted crop, zero outside the frame (KLT.py:65-68), one dword of 4 pixels per thread and step
        const int bx0 = s_box[0], by0 = s_box[2], bdx = s_box[4], bdy = s_box[5];
        const int rw = max(s_box[1] - bx0, 0), rh = max(s_box[3] - by0, 0), rw4 = (rw + 3) >> 2;
        for (int e = tid; e < rw4 * rh; e += NT) {
            const int ry = e / rw4, x4 = (e - ry * rw4) * 4;
            const int sy = by0 + ry + bdy;
            const bool yin = sy >= 0 && sy < io.h;
            uint32_t pack = 0;
            for (int k = 0; k < 4; k++) {
                const int sx = bx0 + x4 + k + bdx;
                const uint32_t v = (yin && x4 + k < rw && sx >= 0 && sx < io.w) ? io.im[(size_t)sy * io.stride + sx] : 0u;
                pack |= v << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(B.warp + (size_t)ry * (size_t)(rw4 * 4) + x4) = pack;  // warp rows are dword padded
        }
    }
}


// ---- stage 2 -> 3: affine (or fallback), remap job, job C (KLT.py:126-133) ---------------------------------------
// ONE thread per stream does the work (the others return): k_klt_glue2 or the tail of k_ransac_fused<2>
__device__ __forceinline__ void klt_glue2_body(StreamWS& ws)
{
    if (threadIdx.x != 0) return;
    const KltIO& io = ws.io;
    const StreamBufs& B = ws.bufs;
    double M[6];
    if (ws.rstatus) {
        for (int k = 0; k < 6; k++) M[k] = ws.M[k];
    } else {
        // "KLT coarse-affine failure" (KLT.py:128-130): the SURF fallback is out of scope -> keep the translation
        ws.flags |= 1;
        M[0] = 1; M[1] = 0; M[2] = ws.t_trans[0]; M[3] = 0; M[4] = 1; M[5] = ws.t_trans[1];
        for (int k = 0; k < 6; k++) ws.M[k] = M[k];
    }
    const float T[6] = {(float)M[0], (float)M[3], (float)M[1], (float)M[4], (float)M[2], (float)M[5]};  // T23.T.astype(float32)
    const int x0 = ws.roi[0], x1 = ws.roi[1], y0 = ws.roi[2], y1 = ws.roi[3];
    const int rw = max(x1 - x0, 0), rh = max(y1 - y0, 0);
    WarpJob& W = ws.warp;
    W.mode = ws.n > 0 ? 1 : -1;
    W.dst = B.warp; W.dst_stride = (rw + 3) & ~3;  // dword rows: the warp kernel stores packed dwords
    for (int k = 0; k < 6; k++) W.T[k] = T[k];
    LKJob& J = ws.lk;
    const vh_lk_params& lk = io.fine;
    fill_pyramid(J.I, io.im0 + (size_t)y0 * io.stride0 + x0, rw, rh, io.stride0, B.roi_lv[0], lk.win, lk.max_level);
    fill_pyramid(J.J, B.warp, rw, rh, (rw + 3) & ~3, B.roi_lv[1], lk.win, lk.max_level);
    ws.pb[0] = PyrBuild{&J.I, 1, 0};
    ws.pb[1] = PyrBuild{&J.J, 1, 0};
    fill_lk_common(J, lk, io.p0, nullptr, ws.n);
    J.order = ws.order;
    J.p_out = io.p_all; J.v_out = io.v;
    J.fbt = io.fbt_fine;
    J.in_scale = 1.f; J.in_off[0] = (float)x0; J.in_off[1] = (float)y0;
    J.out_mode = VH_OUT_AFFINE;
    for (int k = 0; k < 6; k++) J.T[k] = T[k];
    J.stats = &ws.lk_stats[2][0][0];
    if (io.flags) *io.flags = ws.flags;
}

