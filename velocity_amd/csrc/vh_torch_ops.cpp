// PyTorch-ROCm custom-op registration of the hot path (north_star: "exposed to the existing Python driver through PyTorch-ROCm custom ops";
// SURVEY.md section 8b names the ops).  A thin TORCH_LIBRARY(velocity_hip, ...) layer over the C ABI of libvelocity_hip.so
// (include/velocity_hip.h): tensors in, tensors out, launched on torch's CURRENT HIP stream of the inputs' device, no host synchronisation
// (iteration counts / convergence flags come back as device tensors).  No compute lives here -- every op forwards to the same C entry
// point the ctypes binding (velocity_amd/_lib.py) uses, so both bindings give bit-identical results.
//
//   torch.ops.load_library("velocity_amd/libvelocity_torch.so")   # or: import velocity_amd.torch_ops
//   p_all, v, im_small = torch.ops.velocity_hip.klt_main(im, im0, None, p0)
//
// Reference call sites: utils/KLT.py:99-134 (KLTmain), :37-51 (cv2calcOpticalFlowPyrLK), utils/NLS.py:9-33,102-183 (estimateWorldCameraPose ->
// fcnNLS_t / fcnNLS_Rt), utils/common.py:58-64 (world2image), utils/MSV.py:8-49,98-142 (fcnMSV1_t, fcn2vintercept), utils/NLS.py:186-250 (fcnNLS_batch).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "../../include/velocity_hip.h"

#ifndef VH_TORCH_BUILD_ID
#error "build through velocity_amd/_build.py::build_torch_ops (it passes the tree's build id)"
#endif
static const char vh_torch_id[] = "VH_TORCH_BUILD_ID=" VH_TORCH_BUILD_ID;
extern "C" __attribute__((visibility("default"))) const char* vh_torch_build_id(void) { return vh_torch_id + 18; }

namespace {

using at::Tensor;

void vh_check(int rc, const char* what)
{
    TORCH_CHECK(rc == 0, "libvelocity_hip ", what, " failed (rc=", rc, "): ", vh_last_error());
}

// One workspace per (device, stream): the stateless C entry points park their job descriptors in slot 0 of the workspace, so two HIP
// streams must never share one (include/velocity_hip.h, "Conventions").
struct Ctx {
    vh_ctx* h = nullptr;
    int w = 0, hgt = 0, n = 0;
};
std::mutex g_mu;
std::map<std::pair<int, void*>, Ctx> g_ctx;

vh_ctx* workspace(const Tensor& like, int w, int h, int n, void* stream)
{
    std::lock_guard<std::mutex> lock(g_mu);
    Ctx& c = g_ctx[{(int)like.get_device(), stream}];
    if (!c.h || w > c.w || h > c.hgt || n > c.n) {
        const int nw = std::max({w, c.w, 1920}), nh = std::max({h, c.hgt, 1080}), nn = std::max({n, c.n, 8192});
        if (c.h) {
            // kernels of earlier calls may still be reading the old arena
            (void)hipStreamSynchronize((hipStream_t)stream);
            vh_ctx_destroy(c.h);
            c.h = nullptr;
        }
        vh_check(vh_ctx_create(&c.h, 1, nw, nh, nn), "vh_ctx_create");
        c.w = nw; c.hgt = nh; c.n = nn;
    }
    return c.h;
}

void check_image(const Tensor& im, const char* name)
{
    TORCH_CHECK(im.is_cuda() && im.scalar_type() == at::kByte && im.dim() == 2 && im.stride(1) == 1, name, ": expected a CUDA uint8 [H,W] image with unit column stride");
}

Tensor as_points(const Tensor& p, const char* name)
{
    TORCH_CHECK(p.is_cuda() && p.dim() == 2 && p.size(1) == 2, name, ": expected CUDA points [N,2]");
    return p.to(at::kFloat).contiguous();
}

void host_K(const Tensor& K, double out[9])
{
    TORCH_CHECK(K.numel() == 9, "K must hold 9 values (MATLAB row-vector layout [[fx,0,0],[s,fy,0],[cx,cy,1]])");
    Tensor k = K.detach().to(at::kCPU, at::kDouble).contiguous();  // a 72-byte host constant: K.astype(float), utils/NLS.py:22-24,196 (float32 widens exactly)
    for (int i = 0; i < 9; i++) out[i] = k.data_ptr<double>()[i];
}

// ---- KLTmain(im, im0, im0_small, p0) -> (p_all [N,2] f32, v [N] u8, im_small) ; the caller takes p_all[v] (utils/KLT.py:134) -------------------
std::tuple<Tensor, Tensor, Tensor> klt_main(const Tensor& im, const Tensor& im0, const c10::optional<Tensor>& im0_small, const Tensor& p0,
                                            int64_t cw, int64_t cl, int64_t cc, double ce, int64_t fw, int64_t fl, int64_t fc, double fe)
{
    check_image(im, "klt_main(im)");
    check_image(im0, "klt_main(im0)");
    TORCH_CHECK(im.sizes() == im0.sizes(), "klt_main: im and im0 differ in size");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(im.device());
    void* s = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    const int h = (int)im.size(0), w = (int)im.size(1);
    Tensor p = as_points(p0, "klt_main(p0)");
    const int n = (int)p.size(0);
    const int dh = (int)std::lrint(h * 0.25), dw = (int)std::lrint(w * 0.25);
    Tensor small0;
    if (im0_small.has_value() && im0_small->defined()) {
        small0 = im0_small->contiguous();
        TORCH_CHECK(small0.is_cuda() && small0.scalar_type() == at::kByte && small0.size(0) == dh && small0.size(1) == dw, "klt_main: im0_small must be uint8 [round(H/4), round(W/4)]");
    }
    auto opt = im.options();
    Tensor p_all = at::zeros({n, 2}, opt.dtype(at::kFloat)), v = at::zeros({n}, opt.dtype(at::kByte)), small = at::empty({dh, dw}, opt.dtype(at::kByte));
    Tensor flags = at::zeros({1}, opt.dtype(at::kInt));
    vh_lk_params coarse{(int)cw, (int)cl, (int)cc, ce}, fine{(int)fw, (int)fl, (int)fc, fe};
    if (n > 0)
        vh_check(vh_klt_main(workspace(im, w, h, n, s), 0, im.data_ptr<uint8_t>(), im0.data_ptr<uint8_t>(), small0.defined() ? small0.data_ptr<uint8_t>() : nullptr, w, h,
                             (int)im.stride(0), (int)im0.stride(0), p.data_ptr<float>(), n, &coarse, &fine, p_all.data_ptr<float>(), v.data_ptr<uint8_t>(),
                             small.data_ptr<uint8_t>(), flags.data_ptr<int>(), s),
                 "vh_klt_main");
    else
        vh_check(vh_resize_quarter(workspace(im, w, h, 1, s), im.data_ptr<uint8_t>(), w, h, (int)im.stride(0), small.data_ptr<uint8_t>(), s), "vh_resize_quarter");
    return {p_all, v, small};
}

// ---- cv2calcOpticalFlowPyrLK(im1, im2, p1, None, fbt, **lk) -> (p2, status, err, fbe) ; fb_thresh < 0 = no backward pass (utils/KLT.py:37-51) ----
std::tuple<Tensor, Tensor, Tensor, Tensor> pyr_lk(const Tensor& prev, const Tensor& next, const Tensor& pts, int64_t win, int64_t max_level, int64_t max_iter,
                                                  double eps, double fb_thresh)
{
    check_image(prev, "pyr_lk(prev)");
    check_image(next, "pyr_lk(next)");
    TORCH_CHECK(prev.sizes() == next.sizes(), "pyr_lk: images differ in size");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(prev.device());
    void* s = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    const int h = (int)prev.size(0), w = (int)prev.size(1);
    Tensor p = as_points(pts, "pyr_lk(pts)");
    const int n = (int)p.size(0);
    auto opt = prev.options();
    Tensor p2 = at::zeros({n, 2}, opt.dtype(at::kFloat)), v = at::zeros({n}, opt.dtype(at::kByte)), err = at::zeros({n}, opt.dtype(at::kFloat)),
           fbe = at::zeros({n}, opt.dtype(at::kFloat));
    vh_lk_params lk{(int)win, (int)max_level, (int)max_iter, eps};
    if (n > 0)
        vh_check(vh_pyr_lk(workspace(prev, w, h, n, s), prev.data_ptr<uint8_t>(), next.data_ptr<uint8_t>(), w, h, (int)prev.stride(0), (int)next.stride(0), p.data_ptr<float>(), n,
                           &lk, (float)fb_thresh, p2.data_ptr<float>(), v.data_ptr<uint8_t>(), err.data_ptr<float>(), fbe.data_ptr<float>(), s),
                 "vh_pyr_lk");
    return {p2, v, err, fbe};
}

// ---- pose fits (utils/NLS.py:102-183).  x0: host or device, 3 (nls_t) / 6 (nls_rt: rpy, t) values.  info = int32 {iterations, converged} ----------
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> pose(const Tensor& K, const Tensor& p, const Tensor& pw, const Tensor& x0, const Tensor& R, bool findR)
{
    TORCH_CHECK(p.is_cuda() && pw.is_cuda() && p.dim() == 2 && p.size(1) == 2 && pw.dim() == 2 && pw.size(1) == 3 && p.size(0) == pw.size(0),
                "pose: p [n,2] and pw [n,3] must be CUDA tensors with the same number of rows");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(p.device());
    void* s = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    double Kh[9];
    host_K(K, Kh);
    Tensor pf = p.to(at::kFloat).contiguous(), pwd = pw.to(at::kDouble).contiguous();
    Tensor x0h = x0.detach().to(at::kCPU, at::kDouble).contiguous(), Rh = R.detach().to(at::kCPU, at::kDouble).contiguous();
    TORCH_CHECK(x0h.numel() == 6 && Rh.numel() == 9, "pose: x0 must hold [rpy, t] (6 values), R 9 values");
    const int n = (int)pf.size(0);
    auto opt = p.options();
    Tensor t = at::zeros({3}, opt.dtype(at::kFloat)), Rout = at::zeros({3, 3}, opt.dtype(at::kDouble)), res = at::zeros({1}, opt.dtype(at::kDouble)),
           proj = at::zeros({n, 2}, opt.dtype(at::kDouble)), info = at::zeros({2}, opt.dtype(at::kInt));
    vh_check(vh_pose(workspace(p, 0, 0, 0, s), Kh, pf.data_ptr<float>(), pwd.data_ptr<double>(), n, x0h.data_ptr<double>(), Rh.data_ptr<double>(), findR ? 1 : 0,
                     t.data_ptr<float>(), Rout.data_ptr<double>(), res.data_ptr<double>(), proj.data_ptr<double>(), info.data_ptr<int>(), s),
             "vh_pose");
    return {t, Rout, res, proj, info};
}

std::tuple<Tensor, Tensor> nls_t(const Tensor& K, const Tensor& p, const Tensor& pw, const Tensor& x0)
{
    Tensor x = x0.detach().to(at::kCPU, at::kDouble).reshape({-1});
    TORCH_CHECK(x.numel() >= 3, "nls_t: x0 must hold 3 values");
    Tensor x6 = at::cat({at::zeros({3}, x.options()), x.slice(0, 0, 3)});
    auto r = pose(K, p, pw, x6, at::eye(3, x.options()), false);
    return {std::get<0>(r), std::get<4>(r)};
}

std::tuple<Tensor, Tensor, Tensor> nls_rt(const Tensor& K, const Tensor& p, const Tensor& pw, const Tensor& x0)
{
    Tensor x = x0.detach().to(at::kCPU, at::kDouble).reshape({-1});
    auto r = pose(K, p, pw, x, at::eye(3, x.options()), true);
    return {std::get<1>(r).to(at::kFloat), std::get<0>(r), std::get<4>(r)};
}

// estimateWorldCameraPose(K, p, p3, t, R, findR) -> (t f32[3], R f64[3,3], rms residual f64[1], p_proj f64[n,2], info)   (utils/NLS.py:9-33)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> estimate_pose(const Tensor& K, const Tensor& p, const Tensor& p3, const Tensor& x0, const Tensor& R, bool findR)
{
    return pose(K, p, p3, x0, R, findR);
}

// world2image(K, R, t, pw) (utils/common.py:58-64): C = [R; t] @ K is a 12-value host constant
Tensor project(const Tensor& K, const Tensor& R, const Tensor& t, const Tensor& pw)
{
    TORCH_CHECK(pw.is_cuda() && pw.dim() == 2 && pw.size(1) == 3, "project: pw must be CUDA [n,3]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pw.device());
    void* s = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    Tensor Kd = K.detach().to(at::kCPU, at::kDouble).reshape({3, 3}), Rd = R.detach().to(at::kCPU, at::kDouble).reshape({3, 3}),
           td = t.detach().to(at::kCPU, at::kDouble).reshape({1, 3});
    Tensor Cm = at::matmul(at::cat({Rd, td}, 0), Kd).contiguous();
    Tensor pwd = pw.to(at::kDouble).contiguous();
    Tensor out = at::zeros({pwd.size(0), 2}, pwd.options());
    vh_check(vh_world2image(workspace(pw, 0, 0, 0, s), Cm.data_ptr<double>(), pwd.data_ptr<double>(), (int)pwd.size(0), out.data_ptr<double>(), s), "vh_world2image");
    return out;
}

// fcn2vintercept(A [nf,3], U [3,nf,nv]) -> [nv,3]   (utils/MSV.py:98-142)
Tensor two_view_intercept(const Tensor& A, const Tensor& U)
{
    TORCH_CHECK(A.is_cuda() && U.is_cuda() && A.dim() == 2 && A.size(1) == 3 && U.dim() == 3 && U.size(0) == 3 && U.size(1) == A.size(0), "two_view_intercept: A [nf,3], U [3,nf,nv] on the GPU");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(A.device());
    void* s = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    Tensor Ad = A.to(at::kDouble).contiguous(), Ud = U.to(at::kDouble).contiguous();
    Tensor out = at::zeros({Ud.size(2), 3}, Ad.options());
    vh_check(vh_two_view_intercept(workspace(A, 0, 0, 0, s), Ad.data_ptr<double>(), Ud.data_ptr<double>(), (int)Ad.size(0), (int)Ud.size(2), out.data_ptr<double>(), s),
             "vh_two_view_intercept");
    return out;
}

// fcnMSV1_t(K, P [5,N0,nhist] f32, B [nhist,14] f32, ids = nonzero(vg) int32, ii) -> (x f32[3], b0 f64[ng,3], info)   (utils/MSV.py:8-49)
std::tuple<Tensor, Tensor, Tensor> msv1_t(const Tensor& K, const Tensor& P, const Tensor& B, const Tensor& ids, int64_t ii)
{
    TORCH_CHECK(P.is_cuda() && B.is_cuda() && ids.is_cuda() && P.dim() == 3 && P.size(0) == 5 && B.dim() == 2 && B.size(1) == 14 && B.size(0) == P.size(2),
                "msv1_t: P [5,N0,nhist], B [nhist,14], ids [ng] on the GPU");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(P.device());
    void* s = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    double Kh[9];
    host_K(K, Kh);
    Tensor Pf = P.to(at::kFloat).contiguous(), Bf = B.to(at::kFloat).contiguous(), id = ids.to(at::kInt).contiguous();
    const int ng = (int)id.numel(), N0 = (int)Pf.size(1), nh = (int)Pf.size(2);
    auto opt = P.options();
    Tensor U = at::empty({3 * (ii + 1) * std::max(ng, 1)}, opt.dtype(at::kDouble)), x = at::zeros({3}, opt.dtype(at::kFloat)),
           b0 = at::zeros({ng, 3}, opt.dtype(at::kDouble)), info = at::zeros({2}, opt.dtype(at::kInt));
    vh_check(vh_msv1_t(workspace(P, 0, 0, 0, s), Kh, Pf.data_ptr<float>(), Bf.data_ptr<float>(), id.data_ptr<int>(), ng, N0, nh, (int)ii, (K.scalar_type() == at::kFloat && P.scalar_type() == at::kFloat) ? 1 : 0, U.data_ptr<double>(),
                       x.data_ptr<float>(), b0.data_ptr<double>(), info.data_ptr<int>(), s),
             "vh_msv1_t");
    return {x, b0, info};
}

// fcnNLS_batch on packed inputs (utils/NLS.py:198-203: z = [all u | all v] camera-major, x = [points | cam pos | cam rpy]); z / x may carry a leading
// window dimension [nwin, ...] (independent windows solved by one launch sequence).  -> (x, trace [.., max_iter, 2], info [.., 2])
std::tuple<Tensor, Tensor, Tensor> ba_solve(const Tensor& K, const Tensor& z, const Tensor& x0, int64_t nt, int64_t nc, int64_t max_iter)
{
    TORCH_CHECK(z.is_cuda() && x0.is_cuda() && z.dim() == x0.dim() && (z.dim() == 1 || z.dim() == 2), "ba_solve: z and x0 must be CUDA tensors, both [..] or both [nwin, ..]");
    const int64_t nz = 2 * nt * (nc + 1), nx = 3 * nt + 6 * nc;
    TORCH_CHECK(z.size(-1) == nz && x0.size(-1) == nx, "ba_solve: z must hold 2 nt (nc+1) and x0 3 nt + 6 nc values");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(z.device());
    void* s = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    double Kh[9];
    host_K(K, Kh);
    const bool multi = z.dim() == 2;
    const int64_t nw = multi ? z.size(0) : 1;
    TORCH_CHECK(!multi || x0.size(0) == nw, "ba_solve: z and x0 disagree on the number of windows");
    Tensor zd = z.to(at::kDouble).contiguous(), x = x0.to(at::kDouble).contiguous().clone();
    auto opt = z.options();
    const size_t wsb = vh_nls_batch_workspace((int)nt, (int)nc);
    Tensor scratch = at::empty({nw, (int64_t)wsb}, opt.dtype(at::kByte));
    Tensor trace = at::zeros(multi ? std::vector<int64_t>{nw, max_iter, 2} : std::vector<int64_t>{max_iter, 2}, opt.dtype(at::kDouble));
    Tensor info = at::zeros(multi ? std::vector<int64_t>{nw, 2} : std::vector<int64_t>{2}, opt.dtype(at::kInt));
    vh_ctx* c = workspace(z, 0, 0, 0, s);
    if (multi)
        vh_check(vh_nls_batch_multi(c, Kh, zd.data_ptr<double>(), x.data_ptr<double>(), (int)nt, (int)nc, (int)nw, (int)max_iter, trace.data_ptr<double>(), info.data_ptr<int>(),
                                    scratch.data_ptr(), wsb, s),
                 "vh_nls_batch_multi");
    else
        vh_check(vh_nls_batch(c, Kh, zd.data_ptr<double>(), x.data_ptr<double>(), (int)nt, (int)nc, (int)max_iter, trace.data_ptr<double>(), info.data_ptr<int>(),
                              scratch.data_ptr(), wsb, s),
                 "vh_nls_batch");
    return {x, trace, info};
}

}  // namespace

TORCH_LIBRARY(velocity_hip, m)
{
    m.def("klt_main(Tensor im, Tensor im0, Tensor? im0_small, Tensor p0, int coarse_win=15, int coarse_max_level=4, int coarse_max_count=10, float coarse_eps=0.1, "
          "int fine_win=51, int fine_max_level=0, int fine_max_count=30, float fine_eps=0.001) -> (Tensor, Tensor, Tensor)");
    m.def("pyr_lk(Tensor prev, Tensor next, Tensor pts, int win=15, int max_level=4, int max_iter=10, float eps=0.1, float fb_thresh=-1.0) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("nls_t(Tensor K, Tensor p, Tensor pw, Tensor x0) -> (Tensor, Tensor)");
    m.def("nls_rt(Tensor K, Tensor p, Tensor pw, Tensor x0) -> (Tensor, Tensor, Tensor)");
    m.def("estimate_pose(Tensor K, Tensor p, Tensor p3, Tensor x0, Tensor R, bool findR=False) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("project(Tensor K, Tensor R, Tensor t, Tensor pw) -> Tensor");
    m.def("two_view_intercept(Tensor A, Tensor U) -> Tensor");
    m.def("msv1_t(Tensor K, Tensor P, Tensor B, Tensor ids, int ii) -> (Tensor, Tensor, Tensor)");
    m.def("ba_solve(Tensor K, Tensor z, Tensor x0, int nt, int nc, int max_iter=10) -> (Tensor, Tensor, Tensor)");
}

// The inputs that decide the device are CUDA tensors: register under the CUDA (= HIP on ROCm) dispatch key.  There is deliberately NO CPU kernel:
// calling an op with CPU tensors raises (no silent fallback).
TORCH_LIBRARY_IMPL(velocity_hip, CUDA, m)
{
    m.impl("klt_main", &klt_main);
    m.impl("pyr_lk", &pyr_lk);
    m.impl("nls_t", &nls_t);
    m.impl("nls_rt", &nls_rt);
    m.impl("estimate_pose", &estimate_pose);
    m.impl("project", &project);
    m.impl("two_view_intercept", &two_view_intercept);
    m.impl("msv1_t", &msv1_t);
    m.impl("ba_solve", &ba_solve);
}
