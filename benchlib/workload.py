"""The tracker workloads bench.py times: `Workload` (resident streams stepping through a periodic synthetic ring) and `EpisodeWorkload` (short
sequences re-initialised between timed episodes: the hard scene and the reference's real stills).  No oracle import here."""
import ctypes as C
import math
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_RING_CACHE = {}


def make_ring(cfg, ring, device, seed, nsets=1, scene="plane"):
    """nsets x `ring` frames of a periodic plane motion (one texture per set) + the tracks / world points of frame 0.  Rendered once per (size, scene,
    seed, sets): the headline, the reference-parameter leg and the shuffled-track leg walk the same frames (read-only), and rendering 300 float64 1080p
    frames costs more wall time than measuring on them."""
    from velocity_amd import synth

    key = (cfg["w"], cfg["h"], cfg["n"], ring, str(device), seed, nsets, scene)
    if key in _RING_CACHE:
        return _RING_CACHE[key]
    if len(_RING_CACHE) >= 2:  # keep the two newest rings (a C3 ring is 2 GB)
        _RING_CACHE.pop(next(iter(_RING_CACHE)))

    W, H = cfg["w"], cfg["h"]
    K = synth.K_1080P.copy()
    if W != 1920:
        K[:2, :2] *= W / 1920.0
        K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    roll = synth.oscillating_roll(period=float(ring)) if scene == "roll" else None
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=float(ring)), roll=roll)
    frames = torch.stack([synth.render_frame(W, H, m, k, seed=seed + 104729 * t, device=device) for t in range(nsets) for k in range(ring)])
    p0 = synth.grid_tracks(cfg["n"], W, H, seed=(seed & 0xFF) + 1)
    _RING_CACHE[key] = (K, m, frames, p0)
    return K, m, frames, p0



# ----------------------------------------------------------------------------------------------------------------------------------
# the tracker workload
# ----------------------------------------------------------------------------------------------------------------------------------
class Workload:
    """`streams` resident video streams of one config on this rank's GPU: sessions, frame rings, the step loop."""

    def __init__(self, a, cfg, params, scene, streams, steps, warmup, dev, rank, groups=1, host_frames=False):
        from velocity_amd.driver import TrackerSession

        self.a, self.cfg, self.S, self.N, self.W, self.H = a, cfg, streams, cfg["n"], cfg["w"], cfg["h"]
        S, N, W, H = self.S, self.N, self.W, self.H
        lvl = cfg["levels"] - 1 if params == "baseline" else 4
        self.lkc, self.lkf = dict(max_level=lvl), (dict(max_count=a.fine_max_count) if getattr(a, "fine_max_count", 0) > 0 else dict())
        if getattr(a, "coarse_max_count", 0) > 0:
            self.lkc["max_count"] = a.coarse_max_count
        self.params, self.scene, self.ring = params, scene, a.ring
        nhist = min(warmup + steps + 3, 512)
        # one texture set per `ring` streams, so no two resident streams ever work on the same pixels
        nsets = 1 if host_frames else (S + a.ring - 1) // a.ring
        self.K, self.motion, self.frames, self.p0 = make_ring(cfg, a.ring, dev, seed=0xC0FFEE + 7919 * rank, nsets=nsets, scene=scene)
        if getattr(a, "track_order", "raster") == "shuffled":  # the order goodFeaturesToTrack gives (by corner response, i.e. spatially at random)
            self.p0 = self.p0[np.random.default_rng(1234).permutation(N)]
        self.p3 = self.motion.world_points(self.p0)
        self.vp = np.ones(N, bool)  # every valid track takes part in the pose fit (the state after vidExample.py:160)
        G = max(1, min(groups, S))
        assert S % G == 0, "--streams must be a multiple of --groups"
        self.G, self.SG = G, S // G
        SG = self.SG
        self.sessions = [TrackerSession(self.K, W, H, N, nhist=nhist, batch=SG, lk_coarse=self.lkc, lk_fine=self.lkf, msv_frame=0) for _ in range(G)]
        from velocity_amd.driver import session_streams

        self.hip_streams = session_streams(G)  # (a fixed set per process: see driver.session_streams)
        # streams of one texture set share its ring but run at different phases, so every launch sees S different frame pairs
        self.phase = [(7 * b) % a.ring for b in range(S)]
        fset = [0 if host_frames else b // a.ring for b in range(S)]
        self.fset = fset
        if host_frames:
            self.phase = [b % a.ring for b in range(S)]  # consecutive phases: one step's batch is a contiguous slice of the extended host ring
        base_ptr, fbytes = self.frames.data_ptr(), W * H
        for b in range(S):
            self.sessions[b // SG].init_stream(b % SG, self.frames[fset[b] * a.ring + self.phase[b]],
                                               self.motion.apply(self.phase[b], self.p0.astype(float)).astype(np.float32),
                                               self.p3 + self.motion.t(self.phase[b]), self.vp, np.float32([0, 0, 0]))
        tables = torch.empty((a.ring, S), dtype=torch.int64)
        for k in range(a.ring):
            for b in range(S):
                tables[k, b] = base_ptr + (fset[b] * a.ring + (self.phase[b] + k) % a.ring) * fbytes
        self.tables = tables.to(dev)
        self.feeder = None
        if host_frames:
            from velocity_amd.driver import HostFrameFeeder

            assert G == 1, "--host-frames is measured with one session group"
            self.feeder = HostFrameFeeder(S, H, W, depth=3)
            reps = (S + a.ring - 1) // a.ring + 1
            self.host_ring = torch.cat([self.frames[: a.ring].cpu()] * reps, 0)[: a.ring + S].contiguous().pin_memory()  # the decoder's pinned output
        torch.cuda.synchronize()

    def run(self, first, count, ex=None):
        from velocity_amd import _lib as L

        a, G, SG = self.a, self.G, self.SG
        for i in range(first, first + count):
            if self.feeder is not None:
                k = i % a.ring
                s_ = self.feeder.put(self.host_ring[k : k + self.S])  # stream b <- frame (b + i) % ring, straight from pinned memory
                self.sessions[0].step(frames_table=self.feeder.get(s_), time_s=i / 30.0, frame_no=i)
                self.feeder.after_step(s_)
                continue
            row = self.tables[i % a.ring]
            for g in range(G):
                with torch.cuda.stream(self.hip_streams[g]):
                    self.sessions[g].step(frames_table=row[g * SG:(g + 1) * SG], time_s=i / 30.0, frame_no=i)
            if ex is not None and ex.due(i):
                ex.wait()  # stream ordered under RCCL: the previous gather has read `local` before the packs below overwrite it
                for g in range(G):
                    with torch.cuda.stream(self.hip_streams[g]):
                        L.check(self.sessions[g].lib.vh_session_pack_state(self.sessions[g].handle, L.dptr(ex.local[g * SG:(g + 1) * SG]), L.stream_ptr()),
                                "vh_session_pack_state")
                for g in range(1, G):  # side streams: the collective is issued from the current stream, which must see their packs
                    self.hip_streams[0].wait_stream(self.hip_streams[g])
                ex.start()  # no host synchronisation: the collective waits for the current stream itself (dist.TrackStateExchange.start)

    def measure(self, steps, warmup, min_seconds, barrier, reduce_max, ex=None):
        """W warm-up steps, then EXACTLY `steps` timed steps between barrier + synchronize; if that took less than min_seconds, further
        blocks of `steps` steps are timed the same way (all ranks agree on the count) and `value` is computed over all timed steps."""
        from velocity_amd import _lib as L

        ses = self.sessions[0]
        self.run(1, warmup, ex)
        barrier()
        # every stage is timed when the launches are long (many tracks in flight); a latency run (few streams) times its three LK launches only -- an event
        # record between two 5 us kernels is not free
        L.check(ses.lib.vh_profile_detail(ses.ws.handle, 1 if self.N * self.SG >= 3000 else 0), "vh_profile_detail")
        L.check(ses.lib.vh_profile_begin(ses.ws.handle, 32 * steps + 32), "vh_profile_begin")
        barrier()
        t0 = time.perf_counter()
        self.run(1 + warmup, steps, ex)
        if ex is not None:
            ex.wait()
        barrier()
        elapsed = reduce_max(time.perf_counter() - t0)
        prof = dict(ms_sum=(C.c_double * 3)(), launches=(C.c_int * 3)(), iters=(C.c_ulonglong * 3)(), setups=(C.c_ulonglong * 3)())
        L.check(ses.lib.vh_profile_end(ses.ws.handle, prof["ms_sum"], prof["launches"], prof["iters"], prof["setups"]), "vh_profile_end")
        stage_ms, stage_n = (C.c_double * 16)(), (C.c_int * 16)()
        L.check(ses.lib.vh_profile_end_stages(ses.ws.handle, 16, stage_ms, stage_n), "vh_profile_end_stages")
        rois = np.zeros((self.SG, 4), np.int32)
        L.check(ses.lib.vh_klt_rois(ses.ws.handle, rois.ctypes.data_as(L.i32p)), "vh_klt_rois")
        timed, blocks = steps, 1
        if min_seconds > 0 and elapsed < min_seconds:
            more = int(math.ceil((min_seconds - elapsed) / max(elapsed, 1e-6)))
            barrier()
            t0 = time.perf_counter()
            self.run(1 + warmup + steps, more * steps, ex)
            if ex is not None:
                ex.wait()
            barrier()
            elapsed += reduce_max(time.perf_counter() - t0)
            timed += more * steps
            blocks += more
        st = ses.state(0)
        done = warmup + timed
        self.done_steps = done
        truth = self.motion.t((self.phase[0] + done) % self.ring) - self.motion.t(self.phase[0])
        return dict(elapsed=elapsed, timed_steps=timed, blocks=blocks, prof=prof, st=st, alive=st["n_cur"] / self.N, truth=truth,
                    stage_ms=list(stage_ms), stage_n=list(stage_n), rois=rois, lk_kernels=self.lk_kernels(), step_us=1e6 * elapsed / timed)


    def lk_kernels(self):
        """The kernels the library says the three LK launches of the last step took (vh_profile_lk_routes)."""
        from velocity_amd import _lib as L

        ses = self.sessions[0]
        routes, names = (C.c_int * 3)(), C.create_string_buffer(96)
        L.check(ses.lib.vh_profile_lk_routes(ses.ws.handle, routes, names), "vh_profile_lk_routes")
        return [names.raw[32 * k:32 * k + 32].split(b"\0")[0].decode() for k in range(3)]

    def close(self):
        torch.cuda.synchronize()
        self.sessions, self.frames, self.tables, self.feeder = [], None, None, None
        torch.cuda.empty_cache()


class EpisodeWorkload:
    """`streams` resident video streams that each replay SHORT CLIPS, the way the reference is run (vidExample.py: n = 12-20 frames from a start frame):
    every stream is (re)initialised at frame 0 of its clip -- untimed: frame-0 initialisation is not a tracked frame -- then `episode` tracked frames run
    for all streams, timed between synchronisations; repeat.  Tracks therefore die on the gates inside every episode instead of once at the start of a
    long run, which is what keeps the load hard.

      kind "hard_scene":   C2 size (1920 x 1080, 2000 grid tracks, baseline parameters), synth.HardScene -- sensor noise, gain drift, an independently
                           moving foreground, a textureless band, a saturated patch.
      kind "real_texture": the reference's real stills (tests/golden/stills_gray.npz sequence B: 7 frames at 1024 x 768), Harris + cornerSubPix tracks
                           and the plate pose from vh_frame0_init, the reference's own LK parameters and its fcnMSV1_t at frame 5."""

    def __init__(self, kind, a, streams, dev, episode=None, groups=0):
        from velocity_amd import _lib as L
        from velocity_amd import synth
        from velocity_amd.driver import TrackerSession, session_groups

        self.kind, self.a, self.S, self.dev = kind, a, streams, dev
        S = streams
        if kind == "hard_scene":
            W, H, N, ring = 1920, 1080, 2000, a.ring
            self.E = episode or 12
            self.K = synth.K_1080P.copy()
            nsets = (S + ring - 1) // ring
            self.scene = synth.HardScene(self.K, W, H, ring=ring)
            self.frames = torch.stack([self.scene.frame(k, seed=0xC0FFEE + 104729 * t, device=dev) for t in range(nsets) for k in range(ring)])
            p0 = synth.grid_tracks(N, W, H, seed=0xEE + 1)
            # frame-0 state of a clip that starts at ring position f: the grid carried along by the background motion; every world point on the plane
            # (the foreground's tracks are pose outliers, as moving objects are for the reference)
            self.p_ring = torch.from_numpy(np.stack([self.scene.bg.apply(f, p0.astype(float)).astype(np.float32) for f in range(ring)])).to(dev)
            p3 = self.scene.bg.world_points(p0)
            self.p3_ring = torch.from_numpy(np.stack([p3 + self.scene.bg.t(f) for f in range(ring)])).to(dev)
            self.vp = torch.ones(N, dtype=torch.uint8, device=dev)
            self.t0 = np.float32([0, 0, 0])
            self.lkc, self.lkf, self.msv_frame, self.params, self.lk_levels = dict(max_level=2), dict(), 0, "baseline", 2
            self.cfg = dict(w=W, h=H, n=N, levels=3, name="hard scene: 1920 x 1080, 2000 tracks, 3 pyramid levels")
            self.phase = [(7 * b) % ring for b in range(S)]
            self.fset = [b // ring for b in range(S)]
            self.ring = ring
            self.fps = 30.0
        elif kind == "real_texture":
            d = np.load(os.path.join(ROOT, "tests", "golden", "stills_gray.npz"))
            fr, self.times, q, self.K = d["b_frames"], d["b_times"].astype(np.float32), d["b_q"].astype(np.float32), d["b_K"]
            n, H, W = fr.shape
            self.E = n - 1
            copies = min(S, 32)  # streams share 32 copies of the clip: no two of them in flight together read the same addresses more than 8-fold
            self.frames = torch.from_numpy(fr).to(dev).unsqueeze(0).repeat(copies, 1, 1, 1).reshape(copies * n, H, W).contiguous()
            self.ring, self.copies = n, copies
            # frame 0 through the product's own device sequence (vidExample.py:105-127): Harris + cornerSubPix + plate pose + image2world + insidebbox
            ws = L.workspace(W, H, 1100)
            cap = 1004
            p = torch.empty((cap, 2), dtype=torch.float32, device=dev)
            p3 = torch.empty((cap, 3), dtype=torch.float64, device=dev)
            vp = torch.empty(cap, dtype=torch.uint8, device=dev)
            t0, R0 = torch.empty(3, dtype=torch.float32, device=dev), torch.empty(9, dtype=torch.float64, device=dev)
            res0, n0 = torch.empty(1, dtype=torch.float64, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
            from velocity_amd.common import worldPointsLicensePlate

            plate = np.ascontiguousarray(np.asarray(worldPointsLicensePlate("Chile"), np.float64).reshape(12))
            K64, qc = L.host_K(self.K), np.ascontiguousarray(q.reshape(8))
            L.check(ws.lib.vh_frame0_init(ws.handle, L.dptr(self.frames[0]), W, H, W, qc.ctypes.data_as(L.f32p), K64.ctypes.data_as(L.f64p),
                                          plate.ctypes.data_as(L.f64p), 180, 140, 1000, 0.01, 5, 0.04, 5, 100, 0.001, L.dptr(p), L.dptr(p3), L.dptr(vp),
                                          L.dptr(t0), L.dptr(R0), L.dptr(res0), L.dptr(n0), None, L.stream_ptr()), "vh_frame0_init")
            N = int(n0.item())
            self.p_ring, self.p3_ring, self.vp = p[:N].contiguous().unsqueeze(0), p3[:N].contiguous().unsqueeze(0), vp[:N].contiguous()
            self.t0, self.res0 = t0.cpu().numpy(), float(res0.item())
            self.lkc, self.lkf, self.msv_frame, self.params, self.lk_levels = dict(), dict(), 5, "ref", 4
            self.cfg = dict(w=W, h=H, n=N, levels=5, name="the reference's real stills (IMG_4127..4133): 1024 x 768, Harris tracks, utils/KLT.py:106-107")
            self.phase, self.fset = [0] * S, [b % copies for b in range(S)]
            self.fps = None
        else:
            raise ValueError(kind)
        # the streams run as G sessions on G HIP streams like the headline's (driver.session_groups; 0 = that rule)
        G = groups if groups and groups > 0 else session_groups(S, N)
        G = G if S % G == 0 else 1
        self.N, self.W, self.H, self.SG, self.G = N, W, H, S // G, G
        self.sessions = [TrackerSession(self.K, W, H, N, nhist=self.E + 2, batch=self.SG, lk_coarse=self.lkc, lk_fine=self.lkf, msv_frame=self.msv_frame)
                         for _ in range(G)]
        self.session = self.sessions[0]
        from velocity_amd.driver import session_streams

        self.hip_streams = session_streams(G)
        base, fbytes, ring = self.frames.data_ptr(), W * H, self.ring
        # tables[j][b] = pointer to frame (clip start + j) of stream b, for every clip start this workload uses
        self.episodes = 0
        self._tab_cache = {}
        self._base, self._fbytes = base, fbytes
        torch.cuda.synchronize()

    def start_index(self, b, e):
        """Ring position of frame 0 of stream b's clip in episode e."""
        return (self.phase[b] + e * self.E) % self.ring if self.kind == "hard_scene" else 0

    def frame_index(self, b, e, j):
        """Index into self.frames of frame j of stream b's clip in episode e."""
        if self.kind == "hard_scene":
            return self.fset[b] * self.ring + (self.start_index(b, e) + j) % self.ring
        return self.fset[b] * self.ring + j

    def time_of(self, j):
        return float(self.times[j]) if self.kind == "real_texture" else j / 30.0

    def _tables(self, e):
        key = e % (self.ring if self.kind == "hard_scene" else 1)
        if key not in self._tab_cache:
            t = torch.empty((self.E + 1, self.S), dtype=torch.int64)
            for j in range(self.E + 1):
                for b in range(self.S):
                    t[j, b] = self._base + self.frame_index(b, e, j) * self._fbytes
            self._tab_cache[key] = t.to(self.dev)
        return self._tab_cache[key]

    def state(self, b):
        """Host copy of stream b's tracker state (whichever session holds it)."""
        return self.sessions[b // self.SG].state(b % self.SG)

    def init_episode(self, e):
        for b in range(self.S):
            f = self.start_index(b, e) if self.kind == "hard_scene" else 0
            with torch.cuda.stream(self.hip_streams[b // self.SG]):
                self.sessions[b // self.SG].init_stream(b % self.SG, self.frames[self.frame_index(b, e, 0)], self.p_ring[f], self.p3_ring[f], self.vp, self.t0,
                                                        time0=self.time_of(0), res0=getattr(self, "res0", 0.0))

    def run_episode(self, e, sync_each=None):
        """E tracked frames for every stream; returns the wall time between the two synchronisations."""
        tabs = self._tables(e)
        self.init_episode(e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(1, self.E + 1):
            for g in range(self.G):
                with torch.cuda.stream(self.hip_streams[g]):
                    self.sessions[g].step(frames_table=tabs[j][g * self.SG:(g + 1) * self.SG], time_s=self.time_of(j), frame_no=float(j))
            if sync_each is not None:
                torch.cuda.synchronize()
                sync_each(j)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def measure(self, min_seconds=1.0, warm_episodes=1, max_episodes=48):
        from velocity_amd import _lib as L

        ses = self.session
        for e in range(warm_episodes):
            self.run_episode(e)
        L.check(ses.lib.vh_profile_detail(ses.ws.handle, 1 if self.N * self.SG >= 3000 else 0), "vh_profile_detail")
        est = max_episodes * self.E
        L.check(ses.lib.vh_profile_begin(ses.ws.handle, 40 * est + 64), "vh_profile_begin")
        elapsed, eps, e = 0.0, 0, warm_episodes
        while eps < 2 or (elapsed < min_seconds and eps < max_episodes):
            elapsed += self.run_episode(e)
            e += 1
            eps += 1
        self.episodes_done = e
        prof = dict(ms_sum=(C.c_double * 3)(), launches=(C.c_int * 3)(), iters=(C.c_ulonglong * 3)(), setups=(C.c_ulonglong * 3)())
        L.check(ses.lib.vh_profile_end(ses.ws.handle, prof["ms_sum"], prof["launches"], prof["iters"], prof["setups"]), "vh_profile_end")
        stage_ms, stage_n = (C.c_double * 16)(), (C.c_int * 16)()
        L.check(ses.lib.vh_profile_end_stages(ses.ws.handle, 16, stage_ms, stage_n), "vh_profile_end_stages")
        rois = np.zeros((self.SG, 4), np.int32)
        L.check(ses.lib.vh_klt_rois(ses.ws.handle, rois.ctypes.data_as(L.i32p)), "vh_klt_rois")
        # tracks alive per frame of the clip (S column 2 = vg.sum(), vidExample.py:164), mean over a few streams of the last episode
        pick = sorted(set([0, self.S // 3, (2 * self.S) // 3, self.S - 1]))
        tab = np.stack([self.state(b)["S"][: self.E + 1, 2] for b in pick])
        steps = eps * self.E
        routes, names = (C.c_int * 3)(), C.create_string_buffer(96)
        L.check(ses.lib.vh_profile_lk_routes(ses.ws.handle, routes, names), "vh_profile_lk_routes")
        return dict(elapsed=elapsed, timed_steps=steps, episodes=eps, prof=prof, stage_ms=list(stage_ms), stage_n=list(stage_n), rois=rois,
                    alive_by_frame=[round(float(x), 1) for x in tab.mean(0)], streams_sampled=pick, step_us=1e6 * elapsed / steps,
                    lk_kernels=[names.raw[32 * k:32 * k + 32].split(b"\0")[0].decode() for k in range(3)])

    def close(self):
        torch.cuda.synchronize()
        self.session, self.sessions, self.frames, self._tab_cache = None, [], None, {}
        torch.cuda.empty_cache()
