"""Device-resident counterpart of the reference's frame loop (vidExample.py:75-171): TrackerSession keeps the track
state of `batch` video streams on the GPU and advances all of them one frame per step() through vh_session_step.
Inputs are torch CUDA uint8 frames (dense, H x W); results (P, B, S, masks, points) are read back on demand."""
import ctypes as C

import numpy as np

from . import _lib as L


class TrackerSession:
    def __init__(self, K, width, height, n0, nhist=20, batch=1, lk_coarse=None, lk_fine=None, msv_frame=5):
        torch = L.torch_cuda()
        self.torch = torch
        self.batch, self.w, self.h, self.n0, self.nhist = batch, width, height, n0, nhist
        self.ws = L.Workspace(batch, width, height, n0)
        self.lib = self.ws.lib
        self.K64 = L.host_K(K)
        k_is_f32 = int(getattr(K, "dtype", None) == np.float32)  # numpy builds fcnMSV1_t's rays in float32 then (utils/MSV.py:15-17)
        self.lkc = L.lk_params(dict(L.LK_COARSE, **(lk_coarse or {})))
        self.lkf = L.lk_params(dict(L.LK_FINE, **(lk_fine or {})))
        h = C.c_void_p()
        L.check(self.lib.vh_session_create(C.byref(h), self.ws.handle, n0, nhist, width, height, self.K64.ctypes.data_as(L.f64p), k_is_f32,
                                           C.byref(self.lkc), C.byref(self.lkf), int(msv_frame)), "vh_session_create")
        self.handle = h
        self._frames = torch.zeros(batch, dtype=torch.int64, device="cuda")  # device table of frame pointers
        self._keep = [None] * batch
        self._init_keep = []

    def init_stream(self, slot, frame0, p, p3, vp, t0, time0=0.0, frame_no=0.0, res0=0.0):
        """Frame-0 state (vidExample.py:116-131): points p [n0,2], world points p3 [n0,3], pose mask vp, plate pose t0.  numpy arrays or CUDA tensors
        (tensors of the right dtype are used in place: a stream can be re-initialised without touching the host)."""
        torch = self.torch
        f0 = frame0 if isinstance(frame0, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame0))
        f0 = f0.cuda().contiguous()
        assert f0.shape == (self.h, self.w) and f0.dtype == torch.uint8

        def dev(a, np_dtype, t_dtype):
            return L.to_dev(a if isinstance(a, torch.Tensor) else np.asarray(a).astype(np_dtype), t_dtype)

        pd, p3d, vpd = dev(p, np.float32, torch.float32), dev(p3, np.float64, torch.float64), dev(vp, np.uint8, torch.uint8)
        assert pd.shape == (self.n0, 2) and p3d.shape == (self.n0, 3) and vpd.shape == (self.n0,)
        t0 = np.ascontiguousarray(np.asarray(t0, np.float32).reshape(3))
        L.check(self.lib.vh_session_init(self.handle, slot, L.dptr(f0), self.w, L.dptr(pd), L.dptr(p3d), L.dptr(vpd), t0.ctypes.data_as(L.f32p),
                                         float(time0), float(frame_no), float(res0), L.stream_ptr()), "vh_session_init")
        self._keep[slot] = f0
        self._init_keep = [k for k in self._init_keep if k[0] != slot] + [(slot, pd, p3d, vpd)]

    def set_frames(self, frames):
        """frames: list of `batch` CUDA uint8 [H,W] tensors (kept alive until the next call replaces them)."""
        torch = self.torch
        ptrs = []
        for f in frames:
            assert f.is_cuda and f.dtype == torch.uint8 and f.shape == (self.h, self.w) and f.is_contiguous()
            ptrs.append(f.data_ptr())
        self._prev_keep = self._keep
        self._keep = list(frames)
        self._frames.copy_(torch.tensor(ptrs, dtype=torch.int64), non_blocking=False)

    def step(self, frames=None, time_s=0.0, frame_no=0.0, frames_table=None):
        """One frame for every stream.  Either `frames` (list of tensors) or `frames_table` (int64 CUDA tensor of pointers).

        time_s / frame_no: scalars (every stream shares the clock) or sequences / tensors of `batch` values (independent videos,
        each with its own CAP_PROP_POS_MSEC and frame counter).  With `frames_table` the caller owns the frame buffers: the
        frames of step i are read again by step i+1 (as im0) and must stay alive until that step has run."""
        if frames is not None:
            self.set_frames(frames)
        tab = self._frames if frames_table is None else frames_table
        is_t = [hasattr(x, "is_cuda") for x in (time_s, frame_no)]  # torch tensors first: numpy must never see a CUDA tensor
        if not any(is_t) and np.ndim(time_s) == 0 and np.ndim(frame_no) == 0:
            L.check(self.lib.vh_session_step(self.handle, L.dptr(tab), float(time_s), float(frame_no), L.stream_ptr()), "vh_session_step")
            return
        torch = self.torch

        def clock(x, tensor):  # scalar / 0-dim / length-batch, host or device -> float32 CUDA vector of `batch` values
            if tensor:
                t = L.to_dev(x, torch.float32).reshape(-1)
                return t.expand(self.batch).contiguous() if t.numel() == 1 else t
            return L.to_dev(np.array(np.broadcast_to(np.asarray(x, np.float32), (self.batch,))), torch.float32)

        tv, fv = clock(time_s, is_t[0]), clock(frame_no, is_t[1])
        assert tv.numel() == self.batch and fv.numel() == self.batch
        self._clock_keep = (tv, fv)
        L.check(self.lib.vh_session_step_v(self.handle, L.dptr(tab), L.dptr(tv), L.dptr(fv), L.stream_ptr()), "vh_session_step_v")

    def step_bgr(self, frames_bgr, time_s=0.0, frame_no=0.0):
        """One frame for every stream straight from BGR frames (the decoder's output, vidExample.py:89-91): the fused ingest writes the gray frames
        and the quarter-scale images in one pass (vh_session_ingest_bgr), then the step runs on them.  frames_bgr: list of `batch` CUDA uint8 [H,W,3]
        tensors.  The session owns two sets of gray buffers (a frame is read by two steps: as `im`, then as `im0`)."""
        torch = self.torch
        if getattr(self, "_gray", None) is None:
            self._gray = [torch.empty((self.batch, self.h, self.w), dtype=torch.uint8, device="cuda") for _ in range(2)]
            self._gray_tab = [torch.tensor([g[b].data_ptr() for b in range(self.batch)], dtype=torch.int64, device="cuda") for g in self._gray]
            self._gray_i = 0
        ptrs = []
        for f in frames_bgr:
            assert f.is_cuda and f.dtype == torch.uint8 and f.shape == (self.h, self.w, 3) and f.is_contiguous()
            ptrs.append(f.data_ptr())
        self._bgr_keep = list(frames_bgr)
        bgr_tab = torch.tensor(ptrs, dtype=torch.int64).cuda()
        self._bgr_tab_keep = bgr_tab
        k = self._gray_i
        self._gray_i ^= 1
        L.check(self.lib.vh_session_ingest_bgr(self.handle, L.dptr(bgr_tab), 3 * self.w, L.dptr(self._gray_tab[k]), L.stream_ptr()), "vh_session_ingest_bgr")
        self.step(frames_table=self._gray_tab[k], time_s=time_s, frame_no=frame_no)
        return self._gray[k]

    def view(self, slot=0):
        v = L.SessionView()
        L.check(self.lib.vh_session_ptrs(self.handle, slot, C.byref(v)), "vh_session_ptrs")
        return v

    def _rd(self, ptr, count, dtype):
        out = np.empty(count, dtype)
        if count:
            L.check(self.lib.vh_copy_to_host(out.ctypes.data, ptr, out.nbytes, L.stream_ptr()), "vh_copy_to_host")
        return out

    def lk_launches(self):
        """What the library says the three LK launches of the last step were (the launcher's own decisions): kernel names (vh_profile_lk_routes) and launch
        slots per workgroup (vh_profile_lk_tpw) of stage 0 (quarter scale), 1 (coarse ROI), 2 (fine)."""
        routes, names, tpw = (C.c_int * 3)(), C.create_string_buffer(96), (C.c_int * 3)()
        L.check(self.lib.vh_profile_lk_routes(self.ws.handle, routes, names), "vh_profile_lk_routes")
        L.check(self.lib.vh_profile_lk_tpw(self.ws.handle, tpw), "vh_profile_lk_tpw")
        return dict(kernels=[names.raw[32 * k:32 * k + 32].split(b"\0")[0].decode() for k in range(3)], routes=list(routes), slots_per_workgroup=list(tpw))

    def state(self, slot=0):
        """Host copy of the stream state: dict(vg, vp, p, P, B, S, p3, t, res, n_cur, n_pose, frame_i, klt_flags).
        P comes back in the reference's [5, N0, nhist] layout (the device keeps it frame-major, [nhist, 5, N0]: include/velocity_hip.h)."""
        v = self.view(slot)
        n0, nh = self.n0, self.nhist
        n_cur = int(self._rd(v.n_cur, 1, np.int32)[0])
        n_pose = int(self._rd(v.n_pose, 1, np.int32)[0])
        return dict(
            vg=self._rd(v.vg, n0, np.uint8).astype(bool), vp=self._rd(v.vp, n0, np.uint8).astype(bool),
            p=self._rd(v.p, 2 * n_cur, np.float32).reshape(n_cur, 2), ids=self._rd(v.ids, n_cur, np.int32),
            P=np.ascontiguousarray(self._rd(v.P, 5 * n0 * nh, np.float32).reshape(nh, 5, n0).transpose(1, 2, 0)), B=self._rd(v.B, nh * 14, np.float32).reshape(nh, 14),
            S=self._rd(v.S, nh * 9, np.float32).reshape(nh, 9), p3=self._rd(v.p3, 3 * n0, np.float64).reshape(n0, 3),
            t=self._rd(v.t, 3, np.float32), res=float(self._rd(v.res, 1, np.float64)[0]), n_cur=n_cur, n_pose=n_pose,
            frame_i=int(self._rd(v.frame_i, 1, np.int32)[0]), klt_flags=int(self._rd(v.klt_flags, 1, np.int32)[0]),
            pose_info=self._rd(v.pose_info, 2, np.int32))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.torch.cuda.synchronize()
                self.lib.vh_session_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class HostFrameFeeder:
    """Frame ingest from host memory: pinned staging buffers + a side HIP stream, double buffered,
    so the PCIe upload of frame i+1 overlaps the tracking of frame i.  `put(frames)` starts the upload of one frame per
    stream (numpy uint8 [H,W] arrays or a single [batch,H,W] array) and returns the slot; `get(slot)` makes the current
    stream wait for it and returns the device table of frame pointers for TrackerSession.step(frames_table=...);
    `after_step(slot)` hands the previous frame's buffer back.  depth >= 3 keeps the upload one frame ahead."""

    def __init__(self, batch, height, width, depth=3, lanes=None):
        torch = L.torch_cuda()
        self.torch, self.batch, self.depth = torch, batch, depth
        # one upload is split over `lanes` HIP streams (separate DMA engines).  Measured at 64 streams of 1080p (bench.py --host-frames, tools/exp/
        # host_feed_lanes.sh): 1 lane 21.7 k frames/s, 2 lanes 24.6 k (50.8 GB/s of the x16 Gen5 link), 4 lanes 24.1 k, 8 lanes 18.0 k
        import os

        self.lanes = max(1, min(batch, int(os.environ.get("VH_FEEDER_LANES", 0)) or (lanes or (2 if batch >= 8 else 1))))
        self.lane_streams = [torch.cuda.Stream() for _ in range(self.lanes - 1)]
        self.lane_done = [[torch.cuda.Event() for _ in range(self.lanes - 1)] for _ in range(depth)]
        self.pinned = [torch.empty((batch, height, width), dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.dev = [torch.empty((batch, height, width), dtype=torch.uint8, device="cuda") for _ in range(depth)]
        self.tables = [torch.tensor([self.dev[k][b].data_ptr() for b in range(batch)], dtype=torch.int64, device="cuda") for k in range(depth)]
        self.copy_stream = torch.cuda.Stream()
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.free = [torch.cuda.Event() for _ in range(depth)]
        for e in self.free + self.ready:
            e.record()
        self.n = 0
        self._last = None

    def put(self, frames):
        """frames: a pinned torch uint8 tensor [batch,H,W] (uploaded in place, zero staging copies — the decoder should
        write there), or numpy arrays, which are first staged into this feeder's pinned slot."""
        torch = self.torch
        slot = self.n % self.depth
        self.n += 1
        if isinstance(frames, torch.Tensor) and frames.is_pinned():
            src = frames
        else:
            self.ready[slot].synchronize()  # the previous upload out of this staging slot must have finished
            src = self.pinned[slot]
            if isinstance(frames, np.ndarray) and frames.ndim == 3:
                src.numpy()[...] = frames
            else:
                for b, f in enumerate(frames):
                    src[b].numpy()[...] = f
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.free[slot])  # the tracker is done with this device buffer
            if self.lanes == 1:
                self.dev[slot].copy_(src, non_blocking=True)
            else:
                bounds = [round(k * self.batch / self.lanes) for k in range(self.lanes + 1)]
                for k, st in enumerate(self.lane_streams):  # lanes 1.. on their own streams, lane 0 on the copy stream itself
                    a, b = bounds[k + 1], bounds[k + 2]
                    st.wait_event(self.free[slot])
                    with torch.cuda.stream(st):
                        self.dev[slot][a:b].copy_(src[a:b], non_blocking=True)
                        self.lane_done[slot][k].record(st)
                self.dev[slot][: bounds[1]].copy_(src[: bounds[1]], non_blocking=True)
                for ev in self.lane_done[slot]:
                    self.copy_stream.wait_event(ev)
            self.ready[slot].record(self.copy_stream)
        return slot

    def get(self, slot):
        self.torch.cuda.current_stream().wait_event(self.ready[slot])
        return self.tables[slot]

    def after_step(self, slot):
        """Call right after the step that consumed `slot` has been enqueued.  The frame of the PREVIOUS step was that step's
        im0 and is free from here on (a frame is read by two steps: as im1, then as im0)."""
        if self._last is not None:
            self.free[self._last].record(self.torch.cuda.current_stream())
        self._last = slot


# ----------------------------------------------------------------------------------------------------------------------------------
# the reference's driver (vidExample.py:52-178 minus decode and plots): frame-0 initialisation, the frame loop, the table and the summary
# ----------------------------------------------------------------------------------------------------------------------------------
TABLE_HEADER = ("\n" + "%13s" * 9) * 2 % ("image", "procTime", "pointTracks", "metric", "dt", "time", "dx", "distance", "speed",
                                         "#", "(s)", "#", "(pixels)", "(s)", "(s)", "(m)", "(m)", "(km/h)")  # vidExample.py:51-74
ROW_FORMAT = "{:13g}{:13.3f}{:13g}{:13.3f}{:13.3f}{:13.3f}{:13.2f}{:13.2f}{:13.1f}"  # vidExample.py:165


def table_row(S_row):
    """One line of the reference's results table (vidExample.py:164-165) from a 9-column float32 stats row."""
    return ROW_FORMAT.format(*tuple(S_row))


def summary_lines(S, n, frame_numbers, seconds):
    """The closing lines of the reference's run (vidExample.py:177-178)."""
    with np.errstate(all="ignore"):
        a = f"\nSpeed = {S[1:, 8].mean():.2f} +/- {S[1:, 8].std():.2f} km/h\nRes = {S[1:, 3].mean():.3f} pixels"
    b = f"Processed {n:g} images: {np.asarray(frame_numbers)[:]} in {seconds:.2f}s ({n / max(seconds, 1e-12):.2f}fps)\n"
    return [a, b]


def run_sequence(frames, q, K, fps=None, times=None, frame_numbers=None, plate="Chile", roi_border=(700, 500), max_corners=1000, quality=0.01,
                 block=5, harris_k=0.04, subpix=(5, 100, 0.001), msv_frame=5, lk_coarse=None, lk_fine=None, route="session", live=True,
                 out=print, clock=None, name="sequence"):
    """The packaged counterpart of vidExample.py:52-178 (minus video decode and plots) on one clip.

    frames  sequence of n uint8 [H, W] gray frames (numpy arrays or CUDA tensors): what `cv2.cvtColor(cap.read(), BGR2GRAY)` / `cv2.imread(.., 0)` hands
            the reference's loop (vidExample.py:89-93)
    q       float32 [4, 2]: the hand-clicked plate corners of frame 0 (the .mat file's `q`, vidExample.py:31-32)
    K       the camera's 3 x 3 intrinsic matrix, reference layout (images.py:148-151)
    fps / times / frame_numbers   B[i, 12] (seconds; `CAP_PROP_POS_MSEC / 1000` or the EXIF time) and B[i, 13] per frame: either `times` or `fps`
    route   "session": frame 0 through vh_frame0_init (Harris -> cornerSubPix -> plate pose -> image2world -> insidebbox, ONE device sequence) straight
            into a device-resident TrackerSession -- nothing but the frames goes up and nothing but the printed rows comes down (the only route of the
            product; a host loop on the drop-in functions -- what INTEGRATION.md's import switch gives a maintainer -- lives in tools/dropin_loop.py as a
            measurement harness).
    live    True prints every row as its frame finishes (one small read-back per frame, like the reference); False runs the whole clip first.
    out     line sink (default print); clock: time source for the procTime column / fps line (default time.perf_counter).

    Prints the reference's header, one 9-column row per frame (vidExample.py:165) and the `Speed = ... +/- ... km/h / Res = ...` summary (:177-178).
    Returns dict(S, B, P, vg, vp, p, p3, lines, seconds, ms_per_frame, n_tracks0)."""
    import time as _time

    clock = clock or _time.perf_counter
    n = len(frames)
    assert n >= 2, "a clip needs at least two frames"
    q = np.ascontiguousarray(np.asarray(q, np.float32).reshape(4, 2))
    if times is None:
        assert fps, "give `times` or `fps`"
        times = [np.float32(k / fps) for k in range(n)]
    times = [np.float32(t) for t in times]
    frame_numbers = list(range(n)) if frame_numbers is None else list(frame_numbers)
    lines = []

    def emit(line):
        lines.append(line)
        if out is not None:
            out(line)

    emit(f"Starting image processing on {name} ...")  # vidExample.py:50
    emit(TABLE_HEADER)
    t_begin = clock()
    if route == "session":
        res = _run_session(frames, q, K, times, frame_numbers, plate, roi_border, max_corners, quality, block, harris_k, subpix, msv_frame, lk_coarse,
                           lk_fine, emit, clock, live)
    else:
        raise ValueError("route must be 'session' (the host loop on the drop-in functions is a measurement harness: tools/dropin_loop.py::run_sequence_dropin)")
    seconds = clock() - t_begin
    for line in summary_lines(res["S"], n, frame_numbers, seconds):
        emit(line)
    res.update(lines=lines, seconds=seconds, ms_per_frame=1e3 * res.pop("loop_seconds") / (n - 1))
    return res


def _plate_points(country):
    from .common import worldPointsLicensePlate

    return worldPointsLicensePlate(country)


def _run_session(frames, q, K, times, frame_numbers, plate, roi_border, max_corners, quality, block, harris_k, subpix, msv_frame, lk_coarse, lk_fine,
                 emit, clock, live):
    torch = L.torch_cuda()
    tic = clock()
    dev = [f if isinstance(f, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(f)) for f in frames]
    dev = [f.cuda(non_blocking=True).contiguous() for f in dev]
    H, W = dev[0].shape
    n, cap = len(dev), 4 + int(max_corners)
    ses = TrackerSession(K, W, H, cap, nhist=n, batch=1, lk_coarse=lk_coarse, lk_fine=lk_fine, msv_frame=msv_frame)
    lib, ws = ses.lib, ses.ws
    # frame 0 (vidExample.py:105-131): one device sequence; its outputs are the session's frame-0 state without touching the host
    p = torch.empty((cap, 2), dtype=torch.float32, device="cuda")
    p3 = torch.empty((cap, 3), dtype=torch.float64, device="cuda")
    vp = torch.empty(cap, dtype=torch.uint8, device="cuda")
    t0 = torch.empty(3, dtype=torch.float32, device="cuda")
    R0 = torch.empty(9, dtype=torch.float64, device="cuda")
    res0 = torch.empty(1, dtype=torch.float64, device="cuda")
    n0 = torch.empty(1, dtype=torch.int32, device="cuda")
    rois = (C.c_int * 8)()
    plate_w = np.ascontiguousarray(np.asarray(_plate_points(plate), np.float64).reshape(12))
    win, it, eps = subpix
    L.check(lib.vh_frame0_init(ws.handle, L.dptr(dev[0]), W, H, W, q.ctypes.data_as(L.f32p), ses.K64.ctypes.data_as(L.f64p), plate_w.ctypes.data_as(L.f64p),
                               int(roi_border[0]), int(roi_border[1]), int(max_corners), float(quality), int(block), float(harris_k), int(win), int(it),
                               float(eps), L.dptr(p), L.dptr(p3), L.dptr(vp), L.dptr(t0), L.dptr(R0), L.dptr(res0), L.dptr(n0), rois, L.stream_ptr()),
            "vh_frame0_init")
    L.check(lib.vh_session_init_dev(ses.handle, 0, L.dptr(dev[0]), W, L.dptr(p), L.dptr(p3), L.dptr(vp), L.dptr(t0), L.dptr(res0), L.dptr(n0),
                                    float(times[0]), float(frame_numbers[0]), L.stream_ptr()), "vh_session_init_dev")
    ses._keep[0] = dev[0]
    ses._init_keep.append((0, p, p3, vp, t0, res0, n0))
    view = ses.view(0)
    rows = np.zeros((n, 9), np.float32)

    def row(i):
        r = ses._rd(C.c_void_p(view.S + 4 * 9 * i), 9, np.float32)  # one 36-byte read-back (synchronises, like the reference's print)
        return r

    proc = np.zeros(n)
    if live:
        rows[0] = row(0)
        proc[0] = clock() - tic
        rows[0, 1] = proc[0]
        emit(table_row(rows[0]))
    t_loop = clock()
    for i in range(1, n):
        tic = clock()
        ses.step([dev[i]], time_s=float(times[i]), frame_no=float(frame_numbers[i]))
        if live:
            rows[i] = row(i)
            proc[i] = clock() - tic
            rows[i, 1] = proc[i]
            emit(table_row(rows[i]))
    torch.cuda.synchronize()
    loop_seconds = clock() - t_loop
    st = ses.state(0)
    n_tr = int(n0.item())
    if not live:  # the whole clip ran first: every row carries the mean time per frame
        rows = st["S"].copy()
        rows[0, 1] = 0.0
        rows[1:, 1] = loop_seconds / (n - 1)
        for i in range(n):
            emit(table_row(rows[i]))
    S = st["S"].copy()
    S[:, 1] = rows[:, 1]
    k = n_tr  # rows beyond the tracks found at frame 0 never existed (the session was sized for 4 + max_corners)
    return dict(S=S, B=st["B"], P=st["P"][:, :k, :], vg=st["vg"][:k], vp=st["vp"][:k], p=st["p"], p3=st["p3"][:k], ids=st["ids"], n_tracks0=n_tr,
                t0=t0.cpu().numpy(), R0=R0.cpu().numpy().reshape(3, 3), res0=float(res0.item()), boxa=tuple(rois[0:4]), boxb=tuple(rois[4:8]),
                loop_seconds=loop_seconds, klt_flags=st["klt_flags"])


def session_groups(streams, tracks=2000):
    """How many TrackerSessions (each on its own HIP stream) to split `streams` resident video streams of ~`tracks` tracks each into.  The stages of a frame
    step that run ONE workgroup per stream (RANSAC, bookkeeping + pose, the glue kernels) leave the chip nearly idle; with a second session on another HIP
    stream they run while that session's LK launches fill it.  Measured on one MI355X, C2 streams (2000 tracks), frames/s with 1 / 2 / 4 sessions: 2 streams 8.4 /
    8.9 k, 4: 14.0 / 15.1 / 15.0 k, 8: 20.4 / 22.4 / 23.6 k, 16: 28.5 / 30.6 / 31.3 k, 32: 35.0 / 37.7 / 38.1 k, 64: 40.4 / 41.9 / 41.3 k, 128: - / 44.8 / 43.6 k,
    256: 44.5 / 45.7 / 45.7 k; 8 sessions lose everywhere (a session of one or two streams falls back to the one-track-per-wavefront kernels).  A session
    needs enough tracks for the batched kernels: 8 streams of the real stills (278 tracks) LOSE 13 % as four sessions (19.8 -> 17.2 k), so the count is halved
    until a session holds at least 2000 tracks."""
    if streams < 2:
        return 1
    g = 4 if (8 <= streams < 64 and streams % 4 == 0) else (2 if streams % 2 == 0 else 1)
    while g > 1 and (streams // g) * max(int(tracks), 1) < 2000:
        g //= 2
    return g


_SIDE_STREAMS = {}


def session_streams(n):
    """The HIP streams `n` concurrent sessions run on: torch's current stream + n - 1 side streams that are created ONCE per device and handed out again on
    every call.  The runtime multiplexes HIP streams onto a few hardware queues (4 by default, GPU_MAX_HW_QUEUES); a process that keeps creating streams ends
    up with two "concurrent" sessions on one queue -- measured: a two-session leg that ran after a dozen earlier streams had been created fell from 38.7 k to
    36.2 k frames/s, a four-session one from 16.4 k to 11.0 k -- so the side streams are a fixed, small set."""
    torch = L.torch_cuda()
    dev = torch.cuda.current_device()
    pool = _SIDE_STREAMS.setdefault(dev, [])
    while len(pool) < n - 1:
        pool.append(torch.cuda.Stream(device=dev))
    return [torch.cuda.current_stream()] + pool[: max(n - 1, 0)]


def run_sequences(clips, K, plate="Chile", roi_border=(700, 500), max_corners=1000, quality=0.01, block=5, harris_k=0.04, subpix=(5, 100, 0.001),
                  msv_frame=5, lk_coarse=None, lk_fine=None, out=None, sessions=0):
    """Many clips at once: the throughput form of run_sequence.  `clips` = list of dict(frames, q, times[, frame_numbers, name]) of ONE frame size and
    length; every clip is a stream of a device-resident TrackerSession, so a frame step is one launch sequence for all the clips of a session
    (vh_session_step_v: each stream has its own clock).  `sessions`: the clips are split into this many sessions, each on its own HIP stream (0 = auto,
    session_groups(len(clips)): their one-workgroup-per-stream stages overlap the others' LK launches); results do not depend on it.  Frame 0 of every clip
    runs through vh_frame0_init on the device, its outputs feed vh_session_init_dev directly; nothing is read back before the last frame.  Returns one
    result dict per clip (the keys of run_sequence; `lines` = that clip's table and summary, printed through `out` if given), each equal to what
    run_sequence returns for the clip alone."""
    import time as _time

    torch = L.torch_cuda()
    nclip = len(clips)
    assert nclip >= 1
    n = len(clips[0]["frames"])
    dev = [[(f if isinstance(f, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(f))).cuda().contiguous() for f in c["frames"]] for c in clips]
    H, W = dev[0][0].shape
    assert all(len(d) == n and d[0].shape == (H, W) for d in dev), "clips must share frame size and length"
    cap = 4 + int(max_corners)
    G = int(sessions) if sessions and sessions > 0 else session_groups(nclip, max_corners)
    G = max(1, min(G, nclip))
    owner = [b * G // nclip for b in range(nclip)]                    # clip -> session (contiguous blocks)
    members = [[b for b in range(nclip) if owner[b] == g] for g in range(G)]
    slot = {b: members[owner[b]].index(b) for b in range(nclip)}
    hip_streams = session_streams(G)
    main = hip_streams[0]
    for st_ in hip_streams[1:]:
        st_.wait_stream(main)  # the frame uploads above ran on the current stream
    sess = []
    for g in range(G):
        with torch.cuda.stream(hip_streams[g]):  # (a session's context serves one HIP stream: everything of session g is issued on stream g)
            sess.append(TrackerSession(K, W, H, cap, nhist=n, batch=len(members[g]), lk_coarse=lk_coarse, lk_fine=lk_fine, msv_frame=msv_frame))
    plate_w = np.ascontiguousarray(np.asarray(_plate_points(plate), np.float64).reshape(12))
    win, it, eps = subpix
    times = np.stack([np.asarray(c["times"], np.float32) for c in clips])  # [clip, frame]
    fnos = np.stack([np.asarray(c.get("frame_numbers", np.arange(n)), np.float32) for c in clips])
    keep = []
    t_begin = _time.perf_counter()
    for b, c in enumerate(clips):
        ses = sess[owner[b]]
        lib, ws = ses.lib, ses.ws
        with torch.cuda.stream(hip_streams[owner[b]]):
            q = np.ascontiguousarray(np.asarray(c["q"], np.float32).reshape(4, 2))
            bufs = (torch.empty((cap, 2), dtype=torch.float32, device="cuda"), torch.empty((cap, 3), dtype=torch.float64, device="cuda"),
                    torch.empty(cap, dtype=torch.uint8, device="cuda"), torch.empty(3, dtype=torch.float32, device="cuda"),
                    torch.empty(9, dtype=torch.float64, device="cuda"), torch.empty(1, dtype=torch.float64, device="cuda"),
                    torch.empty(1, dtype=torch.int32, device="cuda"))
            p, p3, vp, t0, R0, res0, n0 = bufs
            rois = (C.c_int * 8)()
            L.check(lib.vh_frame0_init(ws.handle, L.dptr(dev[b][0]), W, H, W, q.ctypes.data_as(L.f32p), ses.K64.ctypes.data_as(L.f64p), plate_w.ctypes.data_as(L.f64p),
                                       int(roi_border[0]), int(roi_border[1]), int(max_corners), float(quality), int(block), float(harris_k), int(win), int(it),
                                       float(eps), L.dptr(p), L.dptr(p3), L.dptr(vp), L.dptr(t0), L.dptr(R0), L.dptr(res0), L.dptr(n0), rois, L.stream_ptr()),
                    "vh_frame0_init")
            L.check(lib.vh_session_init_dev(ses.handle, slot[b], L.dptr(dev[b][0]), W, L.dptr(p), L.dptr(p3), L.dptr(vp), L.dptr(t0), L.dptr(res0), L.dptr(n0),
                                            float(times[b, 0]), float(fnos[b, 0]), L.stream_ptr()), "vh_session_init_dev")
            ses._keep[slot[b]] = dev[b][0]
        keep.append((bufs, tuple(rois)))
    t_loop = _time.perf_counter()
    for i in range(1, n):
        for g in range(G):
            with torch.cuda.stream(hip_streams[g]):
                sess[g].step([dev[b][i] for b in members[g]], time_s=times[members[g], i], frame_no=fnos[members[g], i])
    torch.cuda.synchronize()
    loop_seconds = _time.perf_counter() - t_loop
    seconds = _time.perf_counter() - t_begin
    results = []
    for b, c in enumerate(clips):
        st = sess[owner[b]].state(slot[b])
        (p, p3, vp, t0, R0, res0, n0), rois = keep[b]
        k = int(n0.item())
        S = st["S"].copy()
        S[0, 1] = 0.0
        S[1:, 1] = loop_seconds / (n - 1)  # every row carries the mean time of a frame step (of ALL clips)
        lines = [f"Starting image processing on {c.get('name', f'clip {b}')} ...", TABLE_HEADER] + [table_row(S[i]) for i in range(n)]
        lines += summary_lines(S, n, c.get("frame_numbers", list(range(n))), seconds)
        if out is not None:
            for ln in lines:
                out(ln)
        results.append(dict(S=S, B=st["B"], P=st["P"][:, :k, :], vg=st["vg"][:k], vp=st["vp"][:k], p=st["p"], p3=st["p3"][:k], ids=st["ids"], n_tracks0=k,
                            t0=t0.cpu().numpy(), R0=R0.cpu().numpy().reshape(3, 3), res0=float(res0.item()), boxa=rois[0:4], boxb=rois[4:8],
                            klt_flags=st["klt_flags"], lines=lines, seconds=seconds, ms_per_frame=1e3 * loop_seconds / (n - 1), sessions=G))
    return results


def main(argv=None):
    """`python -m velocity_amd.driver clip.npz [--seq b]`: the reference's run (`python vidExample.py`) on a decoded clip.
    clip.npz holds `<seq>_frames` uint8 [n, H, W], `<seq>_times` [n] seconds, `<seq>_q` [4, 2] plate corners and `<seq>_K` [3, 3] (the format of
    tests/golden/stills_gray.npz); decoding videos is out of scope (no decoder in the image)."""
    import argparse

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("clip")
    ap.add_argument("--seq", default="b")
    ap.add_argument("--border", type=int, nargs=2, default=None, help="ROI border around the plate (vidExample.py:108 uses 700 500; the 1024 x 768 stills fixture needs 180 140)")
    ap.add_argument("--msv-frame", type=int, default=5)
    a = ap.parse_args(argv)
    d = np.load(a.clip)
    fr = d[f"{a.seq}_frames"]
    border = tuple(a.border) if a.border else ((700, 500) if fr.shape[2] >= 1900 else (180, 140))
    run_sequence(fr, d[f"{a.seq}_q"], d[f"{a.seq}_K"], times=d[f"{a.seq}_times"], roi_border=border, msv_frame=a.msv_frame,
                 name=f"{a.clip}:{a.seq}")


if __name__ == "__main__":
    main()
