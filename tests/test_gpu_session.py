"""GPU parity of the device-resident frame loop (vidExample.py:133-160 on vh_session_step) vs oracle/session_oracle.py:
bit-exact track-index bookkeeping (vg, vp, compaction, history P rows 0,1,4), <= 1e-4 rel on pose / residual / speed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.session_oracle import SessionOracle  # noqa: E402 (checker only)
from velocity_amd import synth  # noqa: E402


def _scene(W, H, n0, nframes, seed):
    K = synth.K_1080P.copy()
    K[0, 0] = K[1, 1] = 700.0
    K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    m = synth.PlaneMotion(K, z0=3.6)
    frames = [synth.render_frame(W, H, m, k, seed=seed).numpy() for k in range(nframes)]
    p = synth.grid_tracks(n0, W, H, seed=seed & 0xFF)
    # a few hopeless tracks (window far outside the frame) so the masks and the compaction actually change
    p[::37] = np.float32([-40.0, -40.0])
    p3 = m.world_points(p)
    vp = (p[:, 0] > W * 0.3) & (p[:, 0] < W * 0.7) & (p[:, 1] > H * 0.3) & (p[:, 1] < H * 0.7)
    return frames, p, p3, vp, K


@pytest.mark.parametrize("msv_frame", [5, 0])
def test_session_matches_reference_loop(msv_frame):
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, n0, nframes = 640, 360, 400, 9
    frames, p, p3, vp, K = _scene(W, H, n0, nframes, 0xC0FFEE)
    t0 = np.float32([1.5, 0.45, 3.6])
    orc = SessionOracle(K, frames[0], p, p3, vp, t0, time0=0.0, frame_no=0, res0=0.5, nhist=nframes, msv_frame=msv_frame)
    ses = TrackerSession(K, W, H, n0, nhist=nframes, batch=1, msv_frame=msv_frame)
    ses.init_stream(0, frames[0], p, p3, vp, t0, time0=0.0, frame_no=0, res0=0.5)
    for i in range(1, nframes):
        ts = np.float32(i / 29.97)
        orc.step(frames[i], ts, i)
        ses.step([torch.from_numpy(frames[i]).cuda()], time_s=ts, frame_no=i)
        st = ses.state(0)
        assert np.array_equal(st["vg"], orc.vg), f"vg differs at frame {i}"
        assert np.array_equal(st["vp"], orc.vp), f"vp differs at frame {i}"
        assert np.array_equal(st["ids"], np.nonzero(orc.vg)[0])
        assert np.array_equal(st["p"], orc.p), f"compacted points differ at frame {i}"
        np.testing.assert_allclose(st["t"], orc.t, rtol=1e-5)
        np.testing.assert_allclose(st["res"], orc.residuals, rtol=1e-6)
    st = ses.state(0)
    assert st["frame_i"] == nframes - 1
    # history: rows 0,1 (tracked xy) and 4 (frame index) bit-exact incl. the NaN padding; rows 2,3 (projections) to tolerance
    for r in (0, 1, 4):
        assert np.array_equal(st["P"][r], orc.P[r], equal_nan=True)
    assert np.array_equal(np.isnan(st["P"][2:4]), np.isnan(orc.P[2:4]))
    np.testing.assert_allclose(np.nan_to_num(st["P"][2:4]), np.nan_to_num(orc.P[2:4]), rtol=1e-5)
    np.testing.assert_allclose(st["B"], orc.B, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(st["S"][1:, [0, 2, 4, 5]], orc.S[1:, [0, 2, 4, 5]], rtol=0, atol=0)
    np.testing.assert_allclose(st["S"][1:, [3, 6, 7, 8]], orc.S[1:, [3, 6, 7, 8]], rtol=1e-4)  # residual, dx, distance, speed
    np.testing.assert_allclose(st["p3"], orc.p3, rtol=1e-4, atol=1e-5)
    assert st["vg"].sum() < n0  # the hopeless tracks were dropped -> compaction really happened


def test_session_batch_streams_are_independent():
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, n0, nframes, B = 480, 270, 200, 5, 3
    scenes = [_scene(W, H, n0, nframes, 1000 + 77 * b) for b in range(B)]
    t0 = np.float32([1.5, 0.45, 3.6])
    ses = TrackerSession(scenes[0][4], W, H, n0, nhist=nframes, batch=B, msv_frame=0)
    orcs = []
    for b, (frames, p, p3, vp, K) in enumerate(scenes):
        ses.init_stream(b, frames[0], p, p3, vp, t0)
        orcs.append(SessionOracle(K, frames[0], p, p3, vp, t0, nhist=nframes, msv_frame=0))
    for i in range(1, nframes):
        ts = np.float32(i / 30.0)
        ses.step([torch.from_numpy(scenes[b][0][i]).cuda() for b in range(B)], time_s=ts, frame_no=i)
        for b in range(B):
            orcs[b].step(scenes[b][0][i], ts, i)
    for b in range(B):
        st = ses.state(b)
        assert np.array_equal(st["vg"], orcs[b].vg) and np.array_equal(st["vp"], orcs[b].vp)
        assert np.array_equal(st["p"], orcs[b].p)
        np.testing.assert_allclose(st["t"], orcs[b].t, rtol=1e-5)


@pytest.mark.parametrize("B,n0", [(20, 170), (70, 64)])
def test_session_batches_with_partial_stream_sets(B, n0):
    """The LK launches re-index their workgroups by sets of streams (lk_block_xy: 16 for the fine kernel, 64 for the coarse ones; rotated below 64 streams,
    XCD-pinned from 64 on).  20 streams = a full fine set + a tail of 4, one (rotated) coarse set; 70 streams = a pinned coarse set of 64 + a tail of 6.
    Streams carry different scenes and different track counts (every third stream loses half of its tracks before the first step): each stream must
    still equal its own oracle, bit for bit."""
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, nframes = 320, 240, 3
    assert B * n0 >= 3000  # the 4-tracks-per-wavefront coarse kernel (the strip kernel below that is not re-indexed)
    scenes = [_scene(W, H, n0, nframes, 4000 + 31 * b) for b in range(B)]
    t0 = np.float32([1.5, 0.45, 3.6])
    ses = TrackerSession(scenes[0][4], W, H, n0, nhist=nframes, batch=B, msv_frame=0)
    orcs = []
    for b, (frames, p, p3, vp, K) in enumerate(scenes):
        if b % 3 == 1:
            p = p.copy()
            p[n0 // 2:] = np.float32([-60.0, -60.0])  # dies in the first step: this stream's launches are half empty from then on
        ses.init_stream(b, frames[0], p, p3, vp, t0)
        orcs.append(SessionOracle(K, frames[0], p, p3, vp, t0, nhist=nframes, msv_frame=0))
    for i in range(1, nframes):
        ts = np.float32(i / 30.0)
        ses.step([torch.from_numpy(scenes[b][0][i]).cuda() for b in range(B)], time_s=ts, frame_no=i)
        for b in range(B):
            orcs[b].step(scenes[b][0][i], ts, i)
    for b in range(B):
        st = ses.state(b)
        assert np.array_equal(st["vg"], orcs[b].vg) and np.array_equal(st["vp"], orcs[b].vp), b
        assert np.array_equal(st["p"], orcs[b].p), b
        np.testing.assert_allclose(st["t"], orcs[b].t, rtol=1e-5)


@pytest.mark.parametrize("pinned", [False, True])
def test_host_frame_feeder_gives_the_same_tracks(pinned):
    """Frames uploaded through the pinned double-buffered feeder (side stream) == frames already resident in HBM."""
    import torch

    from velocity_amd.driver import HostFrameFeeder, TrackerSession

    W, H, n0, nframes, B = 480, 270, 200, 8, 2
    scenes = [_scene(W, H, n0, nframes, 4242 + 13 * b) for b in range(B)]
    t0 = np.float32([1.5, 0.45, 3.6])
    out = []
    for mode in ("device", "host"):
        ses = TrackerSession(scenes[0][4], W, H, n0, nhist=nframes, batch=B, msv_frame=0)
        for b, (frames, p, p3, vp, K) in enumerate(scenes):
            ses.init_stream(b, frames[0], p, p3, vp, t0)
        feeder = HostFrameFeeder(B, H, W, depth=3) if mode == "host" else None
        held = []
        for i in range(1, nframes):
            batch = np.stack([scenes[b][0][i] for b in range(B)])
            if feeder is None:
                ses.step([torch.from_numpy(batch[b]).cuda() for b in range(B)], time_s=i / 30.0, frame_no=i)
                continue
            src = torch.from_numpy(batch).pin_memory() if pinned else batch
            held.append(src)
            slot = feeder.put(src)
            ses.step(frames_table=feeder.get(slot), time_s=i / 30.0, frame_no=i)
            feeder.after_step(slot)
        out.append([ses.state(b) for b in range(B)])
    for b in range(B):
        for key in ("vg", "vp", "p", "t", "res"):
            assert np.array_equal(out[0][b][key], out[1][b][key]), key


def test_session_survives_total_track_loss():
    """A blank frame kills every track (min-eigenvalue test, KLT.py status): the loop keeps running with zero tracks -- the
    reference would index empty arrays here (resolve by intent, no crash); state stays finite and empty."""
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, n0, nframes = 480, 270, 150, 6
    frames, p, p3, vp, K = _scene(W, H, n0, nframes, 99)
    ses = TrackerSession(K, W, H, n0, nhist=nframes + 2, batch=1, msv_frame=0)
    ses.init_stream(0, frames[0], p, p3, vp, np.float32([1.5, 0.45, 3.6]))
    ses.step([torch.from_numpy(frames[1]).cuda()], time_s=1 / 30.0, frame_no=1)
    assert ses.state(0)["n_cur"] > 100
    blank = torch.full((H, W), 128, dtype=torch.uint8, device="cuda")
    for i in range(2, nframes):
        f = blank if i < 4 else torch.from_numpy(frames[i]).cuda()
        ses.step([f], time_s=i / 30.0, frame_no=i)
        st = ses.state(0)
        assert st["n_cur"] == 0 and st["n_pose"] == 0
        assert not st["vg"].any() and st["p"].shape == (0, 2)
        assert np.all(np.isfinite(st["t"])) and np.isfinite(st["res"])


def test_streams_started_at_different_times_keep_their_own_clock_and_msv_frame():
    """Two streams in one session; stream 1 is RE-INITIALISED with a new clip after 3 steps and gets its own timestamps.  Each stream must equal
    its own reference loop: fcnMSV1_t fires at frame 5 OF EACH CLIP (vidExample.py:155), dt / time / speed come from the stream's own clock."""
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, n0, nfr = 480, 270, 220, 8
    t0 = np.float32([1.5, 0.45, 3.6])
    A = _scene(W, H, n0, nfr + 3, 555)     # stream 0: runs nfr + 2 steps
    B1 = _scene(W, H, n0, 4, 777)          # stream 1, first clip (3 steps)
    B2 = _scene(W, H, n0, nfr, 999)        # stream 1, second clip, started at global step 3
    ses = TrackerSession(A[4], W, H, n0, nhist=nfr + 3, batch=2, msv_frame=5)
    ses.init_stream(0, A[0][0], A[1], A[2], A[3], t0, time0=0.0)
    ses.init_stream(1, B1[0][0], B1[1], B1[2], B1[3], t0, time0=10.0)
    oA = SessionOracle(A[4], A[0][0], A[1], A[2], A[3], t0, time0=0.0, nhist=nfr + 3, msv_frame=5)
    oB = SessionOracle(B1[4], B1[0][0], B1[1], B1[2], B1[3], t0, time0=10.0, nhist=nfr + 3, msv_frame=5)
    kB, clipB = 0, B1
    for g in range(1, nfr + 2):
        if g == 4:  # new clip in slot 1
            ses.init_stream(1, B2[0][0], B2[1], B2[2], B2[3], t0, time0=20.0)
            oB = SessionOracle(B2[4], B2[0][0], B2[1], B2[2], B2[3], t0, time0=20.0, nhist=nfr + 3, msv_frame=5)
            kB, clipB = 0, B2
        kB += 1
        tA, tB = np.float32(g / 29.97), np.float32((20.0 if clipB is B2 else 10.0) + kB / 25.0)
        ses.step([torch.from_numpy(A[0][g]).cuda(), torch.from_numpy(clipB[0][kB]).cuda()], time_s=[tA, tB], frame_no=[g, kB])
        oA.step(A[0][g], tA, g)
        oB.step(clipB[0][kB], tB, kB)
        for slot, orc in ((0, oA), (1, oB)):
            st = ses.state(slot)
            assert st["frame_i"] == orc.i
            assert np.array_equal(st["vg"], orc.vg) and np.array_equal(st["vp"], orc.vp), (g, slot)
            assert np.array_equal(st["p"], orc.p), (g, slot)
            np.testing.assert_allclose(st["p3"], orc.p3, rtol=1e-4, atol=1e-5)
    for slot, orc in ((0, oA), (1, oB)):
        st = ses.state(slot)
        n = orc.i + 1
        np.testing.assert_allclose(st["B"][:n, 12:14], orc.B[:n, 12:14], rtol=0, atol=0)          # the stream's own time / frame number
        np.testing.assert_allclose(st["S"][1:n, [0, 2, 4, 5]], orc.S[1:n, [0, 2, 4, 5]], rtol=0, atol=0)  # i, #tracks, dt, time since start
        np.testing.assert_allclose(st["S"][1:n, [3, 6, 7, 8]], orc.S[1:n, [3, 6, 7, 8]], rtol=1e-4)
    assert oA.i == nfr + 1 and oB.i == nfr - 2 and oB.i >= 5  # both clips passed THEIR frame 5: both re-triangulated


@pytest.mark.parametrize("order", [-1, 1])
def test_session_with_more_than_4096_tracks_takes_the_unfused_path(order):
    """N0 > 4096: the pose fit no longer fits the 256-thread fused frame kernel (bookkeeping, 1024-thread pose, records as three launches).
    order = 1: the LK launches walk the tracks in spatial order (vh_debug_klt_order; the counting sort then takes several passes per thread)."""
    import torch

    from velocity_amd import _lib as L
    from velocity_amd.driver import TrackerSession

    L.load().vh_debug_klt_order(order)

    W, H, n0, nframes = 960, 540, 4500, 4
    frames, p, p3, vp, K = _scene(W, H, n0, nframes, 31337)
    t0 = np.float32([1.5, 0.45, 3.6])
    orc = SessionOracle(K, frames[0], p, p3, vp, t0, nhist=nframes, msv_frame=0)
    ses = TrackerSession(K, W, H, n0, nhist=nframes, batch=1, msv_frame=0)
    ses.init_stream(0, frames[0], p, p3, vp, t0)
    for i in range(1, nframes):
        ts = np.float32(i / 30.0)
        orc.step(frames[i], ts, i)
        ses.step([torch.from_numpy(frames[i]).cuda()], time_s=ts, frame_no=i)
        st = ses.state(0)
        assert np.array_equal(st["vg"], orc.vg) and np.array_equal(st["vp"], orc.vp) and np.array_equal(st["p"], orc.p), i
        np.testing.assert_allclose(st["t"], orc.t, rtol=1e-5)
        np.testing.assert_allclose(st["res"], orc.residuals, rtol=1e-6)
    L.load().vh_debug_klt_order(-1)


def test_session_on_a_rolling_zooming_scene():
    """Camera roll + zoom + translation (SURVEY section 8d's rotation): the stage-3 affine warp leaves its near-identity fast path and gathers;
    tracks, masks and compaction stay bit-identical to the reference loop."""
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, n0, nframes = 800, 450, 500, 7
    K = synth.K_1080P.copy()
    K[0, 0] = K[1, 1] = 900.0
    K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    m = synth.PlaneMotion(K, z0=3.6, traj=lambda k: np.array([0.03 * k, -0.01 * k, 0.12 * k]), roll=lambda k: np.radians(0.9 * k))
    frames = [synth.render_frame(W, H, m, k, seed=2024).numpy() for k in range(nframes)]
    p = synth.grid_tracks(n0, W, H, seed=9)
    p3 = m.world_points(p)
    vp = np.ones(n0, bool)
    t0 = np.float32([0, 0, 0])
    orc = SessionOracle(K, frames[0], p, p3, vp, t0, nhist=nframes, msv_frame=0)
    ses = TrackerSession(K, W, H, n0, nhist=nframes, batch=1, msv_frame=0)
    ses.init_stream(0, frames[0], p, p3, vp, t0)
    for i in range(1, nframes):
        ts = np.float32(i / 30.0)
        orc.step(frames[i], ts, i)
        ses.step([torch.from_numpy(frames[i]).cuda()], time_s=ts, frame_no=i)
        st = ses.state(0)
        assert np.array_equal(st["vg"], orc.vg) and np.array_equal(st["p"], orc.p), i
        np.testing.assert_allclose(st["res"], orc.residuals, rtol=1e-6)
    assert st["vg"].sum() > 0.8 * n0  # the tracker really follows the rotating scene
    truth = m.apply(nframes - 1, p.astype(float))[st["vg"]]
    assert np.median(np.abs(st["p"] - truth)) < 0.1


@pytest.mark.parametrize("params,scene", [("baseline", "plane"), ("ref", "plane"), ("baseline", "roll")])
def test_session_at_the_measured_configuration(params, scene):
    """The configuration bench.py's headline number is measured on (BASELINE config 2): 1080p, 2000 tracks per stream, SEVERAL streams in one
    session, so that >= 3000 tracks are in flight and vh_launch_lk routes to k_lk3<51,1,4> (fine stage, one slot per workgroup) and k_lk_q<15>
    (coarse stages), RANSAC to k_ransac_fused and the bookkeeping + pose to the fused k_sess_frame.  (The 256-stream headline itself takes k_lk_o<15>
    and k_lk3<51,1,4> with FOUR slots per workgroup: test_session_at_the_headline_load_takes_the_headline_routes below.)  Every stream must equal
    its own reference loop: bit-exact vg / vp / ids / p, pose and residual to 1e-5 (north_star: 1e-4)."""
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, N, B, nframes, ring = 1920, 1080, 2000, 2, 7, 60
    K = synth.K_1080P.copy()
    roll = synth.oscillating_roll(period=float(ring)) if scene == "roll" else None
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=float(ring)), roll=roll)
    lkc = dict(max_level=2 if params == "baseline" else 4)
    phases = [0, 7]  # the streams of a bench ring run at different phases of the same motion, with their own textures here
    frames = [[synth.render_frame(W, H, m, ph + k, seed=0xC0FFEE + 104729 * b, device="cuda") for k in range(nframes)] for b, ph in enumerate(phases)]
    p0 = synth.grid_tracks(N, W, H, seed=0xEF)
    p3 = m.world_points(p0)
    vp = np.ones(N, bool)
    ses = TrackerSession(K, W, H, N, nhist=nframes + 1, batch=B, lk_coarse=lkc, lk_fine={}, msv_frame=0)
    orcs = []
    for b, ph in enumerate(phases):
        pb = m.apply(ph, p0.astype(float)).astype(np.float32)
        p3b = p3 + m.t(ph)
        ses.init_stream(b, frames[b][0], pb, p3b, vp, np.float32([0, 0, 0]))
        orcs.append(SessionOracle(K, frames[b][0].cpu().numpy(), pb, p3b, vp, np.float32([0, 0, 0]), nhist=nframes + 1, lk_coarse=lkc, lk_fine={},
                                  msv_frame=0))
    for i in range(1, nframes):
        ts = np.float32(i / 30.0)
        ses.step([frames[b][i] for b in range(B)], time_s=ts, frame_no=i)
        for b in range(B):
            orcs[b].step(frames[b][i].cpu().numpy(), ts, i)
            st = ses.state(b)
            assert np.array_equal(st["vg"], orcs[b].vg) and np.array_equal(st["vp"], orcs[b].vp), (i, b)
            assert np.array_equal(st["ids"], np.nonzero(orcs[b].vg)[0]), (i, b)
            assert np.array_equal(st["p"], orcs[b].p), (i, b)
            np.testing.assert_allclose(st["t"], orcs[b].t, rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(st["res"], orcs[b].residuals, rtol=1e-5)
    for b in range(B):
        st = ses.state(b)
        assert st["n_cur"] > 0.98 * N  # the tracker really follows the scene
        truth = m.t(phases[b] + nframes - 1) - m.t(phases[b])
        assert np.abs(st["t"] - truth).max() < 2e-3
        for r in (0, 1, 4):
            assert np.array_equal(st["P"][r], orcs[b].P[r], equal_nan=True)


def test_session_at_the_headline_load_takes_the_headline_routes():
    """the kernel configuration the headline number times, reached by NATURAL routing (no debug hook): >= 98 304 tracks in flight in
    one launch sequence -- 256 streams x 400 tracks on 640 x 360 frames = 102 400 -- so that vh_lk_route picks k_lk_o<15> for both coarse stages (8 tracks
    per wavefront, spatial launch order) and k_lk3<51, 1, 4> with FOUR launch slots per workgroup for the fine stage, as vh_session_step does at 256 C2
    streams.  The first, a middle and the last stream (each with its own texture and motion phase) are held against SessionOracle for 4 frames: vg / vp /
    ids / p bit for bit, pose and residual to 1e-5; the library's own launch record must name those kernels and slots."""
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, N, B, nframes, ring = 640, 360, 400, 256, 5, 60
    K = synth.K_1080P.copy()
    K[:2, :2] *= W / 1920.0
    K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=float(ring)))
    lkc = dict(max_level=2)
    checked = {0: (0, 11), 131: (9, 22), 255: (31, 33)}  # stream -> (motion phase, texture); every other stream replays one of three filler clips
    filler = [(3, 44), (17, 55), (40, 66)]
    clips = {}

    def clip(ph, tex):
        if (ph, tex) not in clips:
            clips[(ph, tex)] = [synth.render_frame(W, H, m, ph + k, seed=0xC0FFEE + 104729 * tex, device="cuda") for k in range(nframes)]
        return clips[(ph, tex)]

    of = [checked.get(b, filler[b % 3]) for b in range(B)]
    p0 = synth.grid_tracks(N, W, H, seed=0xEF)
    p3 = m.world_points(p0)
    vp = np.ones(N, bool)
    ses = TrackerSession(K, W, H, N, nhist=nframes + 1, batch=B, lk_coarse=lkc, lk_fine={}, msv_frame=0)
    orcs = {}
    for b, (ph, tex) in enumerate(of):
        pb = m.apply(ph, p0.astype(float)).astype(np.float32)
        p3b = p3 + m.t(ph)
        ses.init_stream(b, clip(ph, tex)[0], pb, p3b, vp, np.float32([0, 0, 0]))
        if b in checked:
            orcs[b] = SessionOracle(K, clip(ph, tex)[0].cpu().numpy(), pb, p3b, vp, np.float32([0, 0, 0]), nhist=nframes + 1, lk_coarse=lkc, lk_fine={}, msv_frame=0)
    for i in range(1, nframes):
        ts = np.float32(i / 30.0)
        ses.step([clip(*of[b])[i] for b in range(B)], time_s=ts, frame_no=i)
        rec = ses.lk_launches()
        assert rec["kernels"] == ["k_lk_o<15>", "k_lk_o<15>", "k_lk3<51, 1, 4>"], rec
        assert rec["slots_per_workgroup"] == [1, 1, 4], rec
        for b, o in orcs.items():
            o.step(clip(*of[b])[i].cpu().numpy(), ts, i)
            st = ses.state(b)
            assert np.array_equal(st["vg"], o.vg) and np.array_equal(st["vp"], o.vp), (i, b)
            assert np.array_equal(st["ids"], np.nonzero(o.vg)[0]) and np.array_equal(st["p"], o.p), (i, b)
            np.testing.assert_allclose(st["t"], o.t, rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(st["res"], o.residuals, rtol=1e-5)
    for b in orcs:
        assert ses.state(b)["n_cur"] > 0.9 * N  # the tracker really follows the scene


def test_session_stays_on_the_reference_loop_for_2000_frames():
    """Long-run behaviour (vidExample.py:133-160 iterated): 2000 frames of a periodic scene, no re-detection, so KLT drift accumulates and tracks
    die along the way -- the device session must stay on the reference loop to the end: masks / ids / points bit-exact at every checkpoint,
    the complete history (P rows 0, 1, 4 with their NaN padding), B and S records at the end."""
    import torch

    from velocity_amd.driver import TrackerSession

    W, H, n0, nframes, ring = 480, 270, 150, 2000, 60
    K = synth.K_1080P.copy()
    K[0, 0] = K[1, 1] = 500.0
    K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=float(ring)), roll=synth.oscillating_roll(period=float(ring), max_deg_per_frame=0.03))
    frames = [synth.render_frame(W, H, m, k, seed=77, device="cuda") for k in range(ring)]
    host = [f.cpu().numpy() for f in frames]
    p = synth.grid_tracks(n0, W, H, seed=3, frac=0.9)
    p3 = m.world_points(p)
    vp = np.ones(n0, bool)
    vp[::5] = False
    t0 = np.float32([0, 0, 0])
    orc = SessionOracle(K, host[0], p, p3, vp, t0, nhist=nframes + 1, msv_frame=5)
    ses = TrackerSession(K, W, H, n0, nhist=nframes + 1, batch=1, msv_frame=5)
    ses.init_stream(0, frames[0], p, p3, vp, t0)
    for i in range(1, nframes + 1):
        ts = np.float32(i / 30.0)
        orc.step(host[i % ring], ts, i)
        ses.step([frames[i % ring]], time_s=ts, frame_no=i)
        if i % 100 == 0 or i in (1, 5, 6, 7):
            v = ses.view(0)
            n_cur = int(ses._rd(v.n_cur, 1, np.int32)[0])
            assert n_cur == int(orc.vg.sum()), i
            assert np.array_equal(ses._rd(v.ids, n_cur, np.int32), np.nonzero(orc.vg)[0]), i
            assert np.array_equal(ses._rd(v.p, 2 * n_cur, np.float32).reshape(n_cur, 2), orc.p), i
            np.testing.assert_allclose(ses._rd(v.t, 3, np.float32), orc.t, rtol=1e-5, atol=1e-7)
    st = ses.state(0)
    assert st["frame_i"] == nframes and 0 < st["n_cur"] <= n0
    assert np.array_equal(st["vg"], orc.vg) and np.array_equal(st["vp"], orc.vp)
    for r in (0, 1, 4):
        assert np.array_equal(st["P"][r], orc.P[r], equal_nan=True)
    assert np.array_equal(np.isnan(st["P"][2:4]), np.isnan(orc.P[2:4]))
    np.testing.assert_allclose(np.nan_to_num(st["P"][2:4]), np.nan_to_num(orc.P[2:4]), rtol=1e-5)
    np.testing.assert_allclose(st["B"], orc.B, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st["S"][1:, [0, 2, 4, 5]], orc.S[1:, [0, 2, 4, 5]], rtol=0, atol=0)
    np.testing.assert_allclose(st["S"][1:, [3, 6, 8]], orc.S[1:, [3, 6, 8]], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(st["S"][1:, 7], orc.S[1:, 7], rtol=1e-4)  # accumulated distance (float32 running sum on both sides)
    # the pose the drifting tracks give is still the scene's (what the extras legs of bench.py report as pose_t / pose_t_truth)
    truth = m.t(nframes % ring) - m.t(0)
    assert np.abs(st["t"] - truth).max() < 0.05


def test_session_step_from_bgr_frames_equals_gray_path():
    """TrackerSession.step_bgr (fused ingest: gray + quarter-scale image in one pass, the step skips its own resize) == converting every frame with
    cvtColor first and stepping on the gray frames: identical track state, and both equal the reference loop on the oracle's gray frames."""
    import torch

    from oracle import klt_oracle as KO
    from velocity_amd.driver import TrackerSession

    W, H, n0, nframes, B = 482, 270, 180, 6, 2   # W / 4 = 120.5: the quarter-scale width rounds to even
    rng = np.random.default_rng(5)
    scenes = [_scene(W, H, n0, nframes, 321 + 11 * b) for b in range(B)]
    # colour frames whose luma is NOT the synthetic gray frame (random chroma), so the conversion really matters
    bgr = [[np.clip(np.stack([scenes[b][0][i].astype(int) + rng.integers(-20, 21, (H, W)) * (c - 1) for c in range(3)], -1), 0, 255).astype(np.uint8)
            for i in range(nframes)] for b in range(B)]
    gray = [[KO.bgr2gray(bgr[b][i]) for i in range(nframes)] for b in range(B)]
    t0 = np.float32([1.5, 0.45, 3.6])
    states = []
    for mode in ("gray", "bgr"):
        ses = TrackerSession(scenes[0][4], W, H, n0, nhist=nframes, batch=B, msv_frame=0)
        for b in range(B):
            ses.init_stream(b, gray[b][0], scenes[b][1], scenes[b][2], scenes[b][3], t0)
        for i in range(1, nframes):
            if mode == "gray":
                ses.step([torch.from_numpy(gray[b][i]).cuda() for b in range(B)], time_s=i / 30.0, frame_no=i)
            else:
                out = ses.step_bgr([torch.from_numpy(bgr[b][i]).cuda() for b in range(B)], time_s=i / 30.0, frame_no=i)
                assert np.array_equal(out[0].cpu().numpy(), gray[0][i])
        states.append([ses.state(b) for b in range(B)])
    for b in range(B):
        for key in ("vg", "vp", "p", "ids", "t", "res"):
            assert np.array_equal(states[0][b][key], states[1][b][key]), key
        orc = SessionOracle(scenes[b][4], gray[b][0], scenes[b][1], scenes[b][2], scenes[b][3], t0, nhist=nframes, msv_frame=0)
        for i in range(1, nframes):
            orc.step(gray[b][i], np.float32(i / 30.0), i)
        assert np.array_equal(states[1][b]["vg"], orc.vg) and np.array_equal(states[1][b]["p"], orc.p)


def test_profile_table_overflow_is_an_error_not_an_undercount():
    """a record table sized for fewer launches than a step issues used to drop the later launches silently (under-counted ms / launches).
    Now the dropped records are counted and vh_profile_end / vh_profile_end_stages fail loudly."""
    import ctypes as C

    import torch

    from velocity_amd import _lib as L
    from velocity_amd.driver import TrackerSession

    W, H, n0 = 320, 180, 64
    frames, p, p3, vp, K = _scene(W, H, n0, 3, 5)
    ses = TrackerSession(K, W, H, n0, nhist=6, batch=1, msv_frame=0)
    ses.init_stream(0, frames[0], p, p3, vp, np.float32([1.5, 0.45, 3.6]))
    L.check(ses.lib.vh_profile_detail(ses.ws.handle, 1), "detail")
    L.check(ses.lib.vh_profile_begin(ses.ws.handle, 3), "begin")  # three records: a step issues ~20
    ses.step([torch.from_numpy(frames[1]).cuda()], time_s=1 / 30.0, frame_no=1)
    ms, nl = (C.c_double * 16)(), (C.c_int * 16)()
    rc = ses.lib.vh_profile_end_stages(ses.ws.handle, 16, ms, nl)
    assert rc != 0 and b"too small" in ses.lib.vh_last_error()
    L.check(ses.lib.vh_profile_begin(ses.ws.handle, 64), "begin")  # large enough: fine again
    ses.step([torch.from_numpy(frames[2]).cuda()], time_s=2 / 30.0, frame_no=2)
    L.check(ses.lib.vh_profile_end_stages(ses.ws.handle, 16, ms, nl), "end_stages")
    assert sum(nl) >= 10


def test_session_with_a_never_msv_sentinel_allocates_no_msv_history():
    """msv_frame beyond the history (a 'never' sentinel) must not size the fcnMSV1_t scratch (it used to carve
    24 B x (msv_frame + 1) x n0 per stream): a huge sentinel creates a session as cheaply as msv_frame = 0."""
    from velocity_amd.driver import TrackerSession

    W, H, n0 = 320, 180, 512
    frames, p, p3, vp, K = _scene(W, H, n0, 2, 6)
    ses = TrackerSession(K, W, H, n0, nhist=6, batch=2, msv_frame=1 << 24)  # 24 B * 2^24 * 512 = 206 GB per stream if it were carved
    ses.init_stream(0, frames[0], p, p3, vp, np.float32([1.5, 0.45, 3.6]))
    assert ses.state(0)["n_cur"] == n0


def test_session_rejects_a_reachable_msv_frame_beyond_the_2048_frame_limit():
    """fcnMSV1_t keeps one table entry per history frame in LDS: 2048 frames at most (the reference has no limit).  A session whose history reaches an
    MSV frame beyond that must fail at creation with a message, not silently skip the re-triangulation; an unreachable or disabled MSV frame is fine."""
    from velocity_amd.driver import TrackerSession

    K = synth.K_1080P.copy()
    with pytest.raises(RuntimeError, match="2048"):
        TrackerSession(K, 64, 64, 8, nhist=3000, batch=1, msv_frame=2500)
    TrackerSession(K, 64, 64, 8, nhist=100, batch=1, msv_frame=2500)  # never reached: allowed
    TrackerSession(K, 64, 64, 8, nhist=3000, batch=1, msv_frame=0)     # disabled: allowed
