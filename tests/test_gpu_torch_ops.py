"""torch.ops.velocity_hip.* (TORCH_LIBRARY registration, velocity_amd/csrc/vh_torch_ops.cpp) vs the ctypes shims and the oracle: both bindings
sit on the same C ABI, so their results must be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames():
    from velocity_amd import synth

    W, H, n = 640, 360, 300
    m = synth.AffineMotion(W, H, tx=3.7, ty=-0.8)
    return synth.render_frame(W, H, m, 0).numpy(), synth.render_frame(W, H, m, 1).numpy(), synth.grid_tracks(n, W, H)


def test_klt_main_and_pyr_lk_ops_match_ctypes_and_oracle():
    import torch

    from oracle import klt_oracle as KO
    from velocity_amd import KLT
    from velocity_amd.torch_ops import ops

    f0, f1, p0 = _frames()
    im0, im1, pts = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), torch.from_numpy(p0).cuda()
    p_all, v, small = ops.klt_main(im1, im0, None, pts)
    assert p_all.dtype == torch.float32 and v.dtype == torch.uint8 and small.shape == (90, 160)
    vb = v.bool().cpu().numpy()
    p, cv, csmall = KLT.KLTmain(f1, f0, None, p0)
    ep, ev, esmall = KO.klt_main(f1, f0, None, p0)
    assert np.array_equal(vb, cv) and np.array_equal(p_all.cpu().numpy()[vb], p) and np.array_equal(small.cpu().numpy(), csmall)
    assert np.array_equal(vb, ev) and np.array_equal(p_all.cpu().numpy()[vb], ep) and np.array_equal(small.cpu().numpy(), esmall)
    # second frame with the previous quarter-scale image passed back in (the driver's call pattern, vidExample.py:134)
    p_all2, v2, small2 = ops.klt_main(im0, im1, small, p_all[v.bool()])
    ep2, ev2, _ = KO.klt_main(f0, f1, esmall, ep)
    assert np.array_equal(v2.bool().cpu().numpy(), ev2) and np.array_equal(p_all2.cpu().numpy()[ev2], ep2)
    # pyr_lk with the forward-backward gate
    q, st, err, fbe = ops.pyr_lk(im0, im1, pts, 15, 2, 10, 0.1, 1.0)
    eq, est, eerr, efbe = KO.lk_fb(f0, f1, p0, fbt=1.0, win=15, max_level=2, max_count=10, eps=0.1, return_fbe=True)
    assert np.array_equal(st.bool().cpu().numpy(), est) and np.array_equal(q.cpu().numpy(), eq) and np.array_equal(fbe.cpu().numpy(), efbe)
    # a side stream gets its own workspace and the same answer
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        p_s, v_s, _ = ops.klt_main(im1, im0, None, pts)
    side.synchronize()
    assert torch.equal(p_s, p_all) and torch.equal(v_s, v)


def test_pose_projection_triangulation_ops(golden):
    import torch

    from velocity_amd.NLS import estimateWorldCameraPose, fcnNLS_Rt, fcnNLS_t
    from velocity_amd.torch_ops import ops

    K32 = golden["K32"]
    K = torch.from_numpy(K32)
    for n in (64, 2000):
        p, pw = golden[f"nlst_{n}_p"], golden[f"nlst_{n}_pw"]
        t, info = ops.nls_t(K, torch.from_numpy(p).cuda(), torch.from_numpy(pw).cuda(), torch.tensor([0.0, 0.0, 1.0]))
        assert np.array_equal(t.cpu().numpy(), fcnNLS_t(K32.astype(float), p.astype(float), pw, np.array([0, 0, 1])))
        np.testing.assert_allclose(t.cpu().numpy(), golden[f"nlst_{n}_t"], rtol=2e-6)
        assert int(info[1]) == 1 and 5 <= int(info[0]) <= 30
        tt, R, res, proj, info2 = ops.estimate_pose(K, torch.from_numpy(p).cuda(), torch.from_numpy(pw).cuda(), torch.tensor([0, 0, 0, 0, 0, 1.0]), torch.eye(3), False)
        ct, cR, cres, cproj = estimateWorldCameraPose(K32, p, pw, findR=False)
        assert np.array_equal(tt.cpu().numpy(), ct) and float(res) == cres and np.array_equal(proj.cpu().numpy(), cproj)
        pr = ops.project(K, torch.eye(3), tt.cpu(), torch.from_numpy(pw).cuda())
        np.testing.assert_allclose(pr.cpu().numpy(), cproj, rtol=1e-12)
    p, pw = golden["nlsrt_64_p"], golden["nlsrt_64_pw"]
    R, t, info = ops.nls_rt(K, torch.from_numpy(p).cuda(), torch.from_numpy(pw).cuda(), torch.tensor([0, 0, 0, 0, 0, 1.0]))
    cR, ct = fcnNLS_Rt(K32.astype(float), p.astype(float), pw, np.array([0, 0, 0, 0, 0, 1.0]))
    assert np.array_equal(R.cpu().numpy(), cR) and np.array_equal(t.cpu().numpy(), ct)
    out = ops.two_view_intercept(torch.from_numpy(golden["tri_A"]).cuda(), torch.from_numpy(golden["tri_U"]).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), golden["tri_2v"], rtol=1e-10)
    vg = golden["msv_vg"]
    x, b0, info = ops.msv1_t(K, torch.from_numpy(golden["msv_P"]).cuda(), torch.from_numpy(golden["msv_B"]).cuda(),
                             torch.from_numpy(np.nonzero(vg)[0].astype(np.int32)).cuda(), int(golden["msv_ii"]))
    np.testing.assert_allclose(x.cpu().numpy(), golden["msv_x"], rtol=5e-6)
    np.testing.assert_allclose(b0.cpu().numpy(), golden["msv_b0"], rtol=1e-5, atol=1e-6)


def test_ba_solve_op_single_and_windows(golden, capsys):
    import torch

    from velocity_amd import synth
    from velocity_amd.NLS import fcnNLS_batch
    from velocity_amd.torch_ops import ops

    K32 = golden["K32"]
    scenes = [synth.ba_scene(50, 6, seed=300 + w) for w in range(3)]
    packs = [synth.ba_pack(*s) for s in scenes]
    z = torch.from_numpy(np.stack([p[0] for p in packs])).cuda()
    x0 = torch.from_numpy(np.stack([p[1] for p in packs])).cuda()
    xs, trace, info = ops.ba_solve(torch.from_numpy(K32), z, x0, 50, 5, 10)
    x1, tr1, info1 = ops.ba_solve(torch.from_numpy(K32), z[1], x0[1], 50, 5, 10)
    assert trace.shape == (3, 10, 2) and info.shape == (3, 2) and torch.equal(xs[1], x1) and torch.equal(trace[1], tr1)
    cw, pw, x, tr = fcnNLS_batch(K32, scenes[1][0].copy(), scenes[1][1], scenes[1][2], return_info=True)
    capsys.readouterr()
    assert np.array_equal(x1.cpu().numpy(), x) and np.array_equal(tr1.cpu().numpy(), tr)


def test_ops_have_no_cpu_kernel():
    import torch

    from velocity_amd.torch_ops import ops

    f0, f1, p0 = _frames()
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.klt_main(torch.from_numpy(f1), torch.from_numpy(f0), None, torch.from_numpy(p0))
