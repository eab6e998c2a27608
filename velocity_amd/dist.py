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

    def due(self, frame_index):
        return self.every > 0 and frame_index % self.every == 0

    def start(self):
        """Begin the all-gather of `self.local` (non-blocking)."""
        if self.world == 1 and not dist.is_initialized():
            self.gathered[0].copy_(self.local)
            self._work = None
        else:
            self._work = dist.all_gather_into_tensor(self.gathered.view(-1), self.local.view(-1), group=self.group, async_op=True)
        self.count += 1

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.gathered
