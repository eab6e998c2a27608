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
        """Frame-0 state (vidExample.py:116-131): points p [n0,2], world points p3 [n0,3], pose mask vp, plate pose t0."""
        torch = self.torch
        f0 = frame0 if isinstance(frame0, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame0))
        f0 = f0.cuda().contiguous()
        assert f0.shape == (self.h, self.w) and f0.dtype == torch.uint8
        pd = L.to_dev(np.asarray(p, np.float32), torch.float32)
        p3d = L.to_dev(np.asarray(p3, np.float64), torch.float64)
        vpd = L.to_dev(np.asarray(vp).astype(np.uint8), torch.uint8)
        assert pd.shape == (self.n0, 2) and p3d.shape == (self.n0, 3) and vpd.shape == (self.n0,)
        t0 = np.ascontiguousarray(np.asarray(t0, np.float32).reshape(3))
        L.check(self.lib.vh_session_init(self.handle, slot, L.dptr(f0), self.w, L.dptr(pd), L.dptr(p3d), L.dptr(vpd), t0.ctypes.data_as(L.f32p),
                                         float(time0), float(frame_no), float(res0), L.stream_ptr()), "vh_session_init")
        self._keep[slot] = f0
        self._init_keep.append((pd, p3d, vpd))

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
