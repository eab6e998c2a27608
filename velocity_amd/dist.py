"""Multi-GPU layer: independent video streams shard across ranks (one process per GPU, torch.distributed,
backend "nccl" = RCCL over xGMI); the only exchange on the path is an all-gather of the packed track state every
`every` frames.  Works with any backend (the CPU tests run it over gloo)."""
import numpy as np
import torch
import torch.distributed as dist

HEADER = 8  # words: n_cur, n_pose, frame_i, klt_flags, t[3], res


def record_words(n0):
    return HEADER + 3 * n0


def shard_streams(total_streams, world_size, rank):
    """Contiguous block partition of stream ids over ranks (every rank gets floor or ceil of the mean)."""
    base, extra = divmod(total_streams, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def unpack_state(rec, n0):
    """One packed record (float32 [8 + 3 n0]) -> dict(n_cur, n_pose, frame_i, klt_flags, t, res, p, ids)."""
    rec = rec.detach().cpu().numpy() if isinstance(rec, torch.Tensor) else np.asarray(rec)
    n = int(rec[0])
    ids = rec[HEADER + 2 * n0 : HEADER + 3 * n0].view(np.int32)
    return dict(n_cur=n, n_pose=int(rec[1]), frame_i=int(rec[2]), klt_flags=int(rec[3]), t=rec[4:7].copy(), res=float(rec[7]),
                p=rec[HEADER : HEADER + 2 * n0].reshape(n0, 2)[:n].copy(), ids=ids[:n].copy())


class TrackStateExchange:
    """All-gather of the packed per-stream track state across ranks, issued every `every` frames.

    `local` is a [streams_per_rank, record_words] float32 tensor on the rank's device; gather() returns the
    [world, streams_per_rank, record_words] tensor of every rank's state.  The message is tiny (24 KB per stream at
    n0 = 2000), i.e. latency bound on xGMI, so it is issued asynchronously and only waited for when it is consumed.
    """

    def __init__(self, streams_per_rank, n0, every=30, device=None, group=None):
        self.group, self.every, self.n0 = group, every, n0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.local = torch.zeros((streams_per_rank, record_words(n0)), dtype=torch.float32, device=dev)
        self.gathered = torch.zeros((self.world, streams_per_rank, record_words(n0)), dtype=torch.float32, device=dev)
        self._work = None
        self.count = 0
        self.host_seconds = 0.0  # host time spent inside start() / wait() (under RCCL: enqueue cost only, the collective is stream ordered)
        # gloo (tests only) has no device all-gather: stage through host buffers there; RCCL gathers device to device
        self._host = dist.is_initialized() and dist.get_backend(group) == "gloo" and self.local.is_cuda
        if self._host:
            self._h_local = torch.zeros_like(self.local, device="cpu")
            self._h_gathered = torch.zeros_like(self.gathered, device="cpu")

    def due(self, frame_index):
        return self.every > 0 and frame_index % self.every == 0

    def start(self):
        """Begin the all-gather of `self.local` (non-blocking).  Stream ordered: whatever wrote `local` on the CURRENT stream (vh_session_pack_state)
        is waited for by the collective itself (RCCL: event on the current stream; gloo: the staging copy runs on it) -- no host synchronisation."""
        import time

        t0 = time.perf_counter()
        self._start()
        self.host_seconds += time.perf_counter() - t0

    def _start(self):
        if self.world == 1 and not dist.is_initialized():
            self.gathered[0].copy_(self.local)
            self._work = None
        elif self._host:
            self._h_local.copy_(self.local)
            self._work = dist.all_gather_into_tensor(self._h_gathered.view(-1), self._h_local.view(-1), group=self.group, async_op=True)
        else:
            self._work = dist.all_gather_into_tensor(self.gathered.view(-1), self.local.view(-1), group=self.group, async_op=True)
        self.count += 1

    def wait(self):
        import time

        t0 = time.perf_counter()
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._host:
                self.gathered.copy_(self._h_gathered)
        self.host_seconds += time.perf_counter() - t0
        return self.gathered

    def describe(self):
        """What the process group really is, for the bench record: backend, world size and every rank's device (all-gathered)."""
        if not dist.is_initialized():
            return dict(backend=None, world_size=1, devices=[], exchanges=self.count, exchange_host_ms_total=round(1e3 * self.host_seconds, 3))
        me = dict(rank=dist.get_rank(self.group))
        if torch.cuda.is_available():
            d = torch.cuda.current_device()
            pr = torch.cuda.get_device_properties(d)
            me.update(device=d, name=pr.name, pci_bus_id=getattr(pr, "pci_bus_id", None), gcn_arch=getattr(pr, "gcnArchName", None))
        devs = [None] * self.world
        dist.all_gather_object(devs, me, group=self.group)
        return dict(backend=dist.get_backend(self.group), world_size=self.world, devices=devs, exchanges=self.count,
                    exchange_host_ms_total=round(1e3 * self.host_seconds, 3), bytes_per_rank_per_exchange=int(self.local.numel() * 4))

    def measure_idle_latency(self):
        """Device latency of ONE more all-gather of the same size on an otherwise idle device, bracketed by events on the current stream.  Collective:
        every rank calls it.  It runs on scratch buffers of its own -- `gathered`, `count` and `host_seconds` of the timed run are left untouched -- and
        refuses to run while an exchange is in flight.  Returns microseconds, or None where a device-side figure means nothing (gloo stages the copy on
        the host and its wait blocks the host; no process group; CPU tensors)."""
        if self._work is not None:
            raise RuntimeError("TrackStateExchange.measure_idle_latency: an exchange is still in flight (call wait() first)")
        if not dist.is_initialized() or self._host or not self.local.is_cuda:
            return None
        src = torch.zeros_like(self.local)
        dst = torch.zeros_like(self.gathered)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.all_gather_into_tensor(dst.view(-1), src.view(-1), group=self.group)  # warm-up (communicator set-up is not latency)
        torch.cuda.synchronize()
        e0.record()
        dist.all_gather_into_tensor(dst.view(-1), src.view(-1), group=self.group)
        e1.record()
        torch.cuda.synchronize()
        return round(1e3 * e0.elapsed_time(e1), 1)


# ----------------------------------------------------------------------------------------------------------------
# multi-GPU bundle adjustment (BA row): tie points are sharded over ranks, the nc free cameras are
# replicated; per LM iteration ONE all-reduce of the reduced camera system [S | rhs | sums] (6nc x (6nc+1) + 4 float64,
# 104 KB at nc = 19) and one of 4 scalars.  No host synchronisation inside the loop.
# ----------------------------------------------------------------------------------------------------------------
def shard_tracks(nt, world_size, rank):
    """Contiguous block of tie-point indices owned by `rank`."""
    ids = shard_streams(nt, world_size, rank)
    return ids[0] if ids else 0, (ids[-1] + 1) if ids else 0


def fcnNLS_batch_sharded(K, P, pw, cw, max_iter=10, group=None, timing=None):
    """fcnNLS_batch (utils/NLS.py:186-250) with the tie points sharded over the ranks of `group`.

    Every rank passes the FULL P / pw / cw (like the reference call) and gets the full (cw, pw) back; internally it
    packs and solves only its own block of tracks.  Works with any world size, including an uninitialised
    process group (single rank).  Returns (cw, pw, trace) with trace = [(rms residual, rms delta)] per iteration.
    `timing` (a dict) receives "loop_ms": the device time of the LM iterations alone (HIP events around the loop: phases + all-reduces,
    without the host-side packing before and the gather after).
    """
    import ctypes as C

    from . import _lib as L

    tc = L.torch_cuda()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    collective = dist.is_initialized()  # also with one rank: the same stream-ordered RCCL calls as a multi-GPU run
    P = np.asarray(P)
    pw = np.asarray(pw, np.float64)
    cw = np.asarray(cw, np.float64)
    keep = np.isfinite(P[4]).sum(1) == P.shape[2]  # NLS.py:190
    P, pw = P[:, keep], pw[keep]
    _, nt_total, nf = P.shape
    nc = nf - 1
    lo, hi = shard_tracks(nt_total, world, rank)
    nt = hi - lo
    if nt < 1:
        raise ValueError("more ranks than tie points")
    Pl = P[:, lo:hi]
    z = np.concatenate([Pl[0].T.reshape(-1), Pl[1].T.reshape(-1)]).astype(np.float64)  # NLS.py:198-199 on the local tracks
    x0 = np.concatenate((pw[lo:hi], cw[1:], np.zeros((nc, 3)))).reshape(-1)
    K64 = L.host_K(K)
    zd = L.to_dev(z, tc.float64)
    xd = L.to_dev(x0, tc.float64).clone()
    trace = tc.zeros((max_iter, 2), dtype=tc.float64, device="cuda")
    info = tc.zeros(2, dtype=tc.int32, device="cuda")
    ws = L.workspace()
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = tc.empty(nbytes // 8 + 1, dtype=tc.float64, device="cuda")  # float64 so that the exchange span is a view of it
    import os

    if os.environ.get("VH_POISON_WORKSPACE"):
        scratch.fill_(float("nan"))
    off, cnt = C.c_size_t(), C.c_size_t()

    def phase(ph, it=0):
        L.check(ws.lib.vh_nls_batch_phase(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, nt_total, int(rank == 0), ph, it,
                                          L.dptr(trace), L.dptr(info), L.dptr(scratch), nbytes, C.byref(off), C.byref(cnt), L.stream_ptr()),
                "vh_nls_batch_phase")

    phase(0)
    assert off.value % 8 == 0
    span = scratch[off.value // 8 : off.value // 8 + cnt.value]
    # the span ends with the 4 accumulators [sum r^2, sum delta^2, -, -]: sum r^2 travels with the first all-reduce (it is final
    # after phase 1), so the second one carries ONLY sum delta^2 -- reducing the whole tail again would count sum r^2 world times
    sum_delta = span[-3:-2]
    if timing is not None:
        ev0, ev1 = tc.cuda.Event(enable_timing=True), tc.cuda.Event(enable_timing=True)
        ev0.record()
    dbg_sync = bool(os.environ.get("VH_AR_SYNC"))
    for it in range(max_iter):
        phase(1, it)
        if collective:
            if dbg_sync: tc.cuda.synchronize()
            dist.all_reduce(span, group=group)
            if dbg_sync: tc.cuda.synchronize()
        phase(2, it)
        if collective:
            if dbg_sync: tc.cuda.synchronize()
            dist.all_reduce(sum_delta, group=group)
            if dbg_sync: tc.cuda.synchronize()
        phase(3, it)
    if timing is not None:
        ev1.record()
        ev1.synchronize()
        timing["loop_ms"] = ev0.elapsed_time(ev1)
    info_h = info.cpu().numpy()
    x = xd.cpu().numpy()
    # gather the point blocks (cameras are identical on every rank)
    if collective:
        # equal-sized padded blocks: gloo (CPU tests, one-device GPU tests) cannot gather ragged lists
        sizes = [shard_tracks(nt_total, world, r) for r in range(world)]
        pad = max(b - a for a, b in sizes)
        mine = tc.zeros((pad, 3), dtype=tc.float64, device="cuda")
        mine[:nt] = xd[: 3 * nt].view(nt, 3)
        parts = [tc.zeros((pad, 3), dtype=tc.float64, device="cuda") for _ in sizes]
        dist.all_gather(parts, mine, group=group)
        pw_out = tc.cat([q[: b - a] for q, (a, b) in zip(parts, sizes)]).cpu().numpy()
    else:
        pw_out = x[: 3 * nt].reshape(nt, 3).copy()
    cw_out = np.concatenate((np.zeros((1, 3)), x[3 * nt : 3 * nt + 3 * nc].reshape(nc, 3)), 0)
    return cw_out, pw_out, trace.cpu().numpy()[: info_h[0]]
