"""Multi-GPU plumbing for frame-sharded motion estimation (one process per GPU).

The path shards by FRAME: an ME job depends only on (current plane, reference plane(s), job
descriptor), so the frames of a lookahead / GOP batch are dealt to the ranks and searched
independently.  The only exchange step is the broadcast of each reconstructed reference plane
from the rank that owns it to the others (RCCL over xGMI on MI355X; gloo on CPU in the tests).
Results (8 bytes per PU) stay with the rank that produced them, or are gathered to one rank.

The reference encoder has no counterpart (it is single-process, host threads only); the hook
where a reference plane becomes valid is its row-publish in source/encoder/framefilter.cpp:257.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def single_rank_group(device=None) -> str:
    """A process group of ONE rank (RCCL on `device` when it initialises, else gloo), so that the collectives of the
    multi-GPU path really run on a single GPU; returns the backend name.  Does nothing if a group exists."""
    if dist.is_initialized():
        return dist.get_backend()
    last = None
    for backend in (("nccl", "gloo") if device is not None else ("gloo",)):
        try:
            kw = {"device_id": device} if backend == "nccl" else {}
            # (an in-process store: no port, and nothing a launcher put into the environment - its rendezvous store, say - is
            # consulted; a tcp:// rendezvous under torch.distributed.run would wait as a client for a store nobody started)
            dist.init_process_group(backend, store=dist.HashStore(), rank=0, world_size=1, **kw)
            return backend
        except Exception as e:      # pragma: no cover - depends on the box
            last = e
    raise last


def shard_frames(n_frames: int, rank_: int, world_: int) -> List[int]:
    """Frames of a batch owned by `rank_`: frame i -> rank i % world (frame i of an 8-frame
    lookahead batch runs on GPU i when world == 8)."""
    return list(range(rank_, n_frames, world_))


def broadcast_plane(plane: torch.Tensor, src: int = 0, engine=None) -> torch.Tensor:
    """Broadcast a padded reference plane (uint8/uint16 tensor, same shape on every rank) from
    `src` to all ranks, in place.  One collective per plane: the planes are small (2.6 MB at
    1080p, 9.3 MB at 4K) and xGMI is point-to-point, so a single direct fan-out beats any
    per-row or ring schedule.  If `engine` is given the copy is ordered against the engine's stream
    on the GPU: it starts after the searches queued so far (they may still read the plane), and work
    queued on the engine afterwards starts after it.  The host does not wait.  (PlaneRing overlaps
    the copy with the searches of the previous frame.)"""
    grouped = dist.is_available() and dist.is_initialized()
    ordered = engine is not None and plane.is_cuda and grouped
    if ordered:
        cur = torch.cuda.current_stream(plane.device)
        cur.wait_stream(engine.torch_stream())
    if grouped:
        dist.broadcast(plane, src=src)
    if ordered:
        engine.torch_stream().wait_stream(cur)
    return plane


def broadcast_plane_rows(plane: torch.Tensor, first_row: int, n_rows: int, src: int = 0, engine=None) -> torch.Tensor:
    """The same for rows [first_row, first_row + n_rows) of the PADDED plane tensor only (a contiguous
    slice): the exchange step of row-wise publishing, where the owner of a reconstructed picture hands
    out each CTU row (with its extended borders) as soon as it is final."""
    broadcast_plane(plane[first_row:first_row + n_rows], src=src, engine=engine)
    return plane


class PlaneRing:
    """Reference planes arriving one per frame, each broadcast from its owner into one of two
    buffers.  Per frame k the caller runs
        b = ring.acquire(k)        # buffer holding plane k (broadcast by prefetch(k), or now)
        ... queue the search of frame k on the engine, reading buffer b ...
        ring.prefetch(k + 1)       # start the next broadcast into the other buffer, beside the search
        ring.release(k)            # mark the end of the searches reading buffer b
    On the GPU every hand-over is an event between two streams - the collective runs on / is joined to
    torch's current stream, the searches on the engine's stream - and the host never waits:
      * acquire / prefetch: the broadcast into a buffer starts after the event that release() recorded
        behind the last search reading that buffer (two frames earlier); the engine's stream then
        waits for the broadcast's completion event before anything queued later runs;
      * release(k): records "buffer k & 1 is free again" on the engine's stream.
    So broadcast k + 1 overlaps search k, and a sequence of steps queues up without host round trips
    (the caller synchronises once at the end).  With no process group nothing is copied and the ring
    only hands out buffer indices; with a process group of one rank the collective still runs (it moves
    nothing), which is how a single GPU exercises this path.

    The OWNER of the planes (rank `src`) refills a buffer with plane k before its broadcast.  That write must
    not start while searches of frame k - 2 still read the buffer: call publish(k, tensor) (or wait_free(k) and
    write the buffer yourself, on torch's current stream) - never write buffers[slot(k)] directly.

    overlap=False (the default) keeps the broadcast of frame k + 1 out of prefetch(): it is then queued in
    acquire(k + 1) (serial, still without host waits).  overlap=True queues it beside search k; that mode has
    run on one GPU with a one-rank group only, so it is opt-in.

    On the CPU (gloo tests) the collective is synchronous and there is nothing to order."""

    def __init__(self, buffers, engine=None, src: int = 0, overlap: bool = False, collective=None):
        assert len(buffers) == 2
        self.buffers = list(buffers)
        self.engine = engine
        self.src = src
        self.overlap = overlap
        self.gpu = engine is not None and self.buffers[0].is_cuda
        self.ready = -1                     # index of the newest plane already broadcast
        # (collective=False: hand out buffers without broadcasting although a process group exists)
        self.collective = (dist.is_available() and dist.is_initialized()) if collective is None else bool(collective)
        self.free = [None, None]            # per buffer: event behind the last search that read it
        self.filled = [None, None]          # per buffer: event behind the broadcast that filled it
        if self.gpu:
            self.free_pool = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(2)]
            self.fill_pool = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(2)]
            self.uses = [0, 0]

    def slot(self, k: int) -> int:
        return k & 1

    def wait_free(self, k: int, stream=None) -> None:
        """Order `stream` (default: torch's current stream) behind the searches that last read the buffer
        plane k goes into (frame k - 2's).  The owner calls this before it writes plane k into that buffer."""
        b = self.slot(k)
        if self.gpu and self.free[b] is not None:
            st = stream if stream is not None else torch.cuda.current_stream(self.buffers[b].device)
            st.wait_event(self.free[b])

    def publish(self, k: int, plane: torch.Tensor) -> int:
        """Owner side: copy plane k (a host or device tensor of the buffers' shape) into its ring buffer once
        the searches still reading that buffer are done, on torch's current stream (the stream the broadcast is
        queued on, so the broadcast follows the copy).  Ranks other than `src` do nothing.  Returns the slot."""
        b = self.slot(k)
        if rank() == self.src:
            self.wait_free(k)
            self.buffers[b].copy_(plane, non_blocking=True)
        return b

    def _broadcast(self, k: int) -> None:
        b = self.slot(k)
        if self.gpu:
            cur = torch.cuda.current_stream(self.buffers[b].device)
            if self.free[b] is not None:
                cur.wait_event(self.free[b])            # the searches of frame k - 2 have read the buffer
            dist.broadcast(self.buffers[b], src=self.src)
            ev = self.fill_pool[b][self.uses[b] & 1]
            ev.record(cur)
            self.filled[b] = ev
        else:
            dist.broadcast(self.buffers[b], src=self.src)
        self.ready = k

    def acquire(self, k: int, stream_set: int = 0) -> int:
        """stream_set: the engine's stream set frame k's searches are queued on (frames in flight, one set each)"""
        b = self.slot(k)
        if self.collective:
            if self.ready < k:
                self._broadcast(k)
            if self.gpu and self.filled[b] is not None:
                self.engine.wait_event(self.filled[b], stream_set)  # searches queued on that set from now on see plane k
                self.filled[b] = None
        return b

    def prefetch(self, k: int) -> None:
        if self.overlap and self.collective and self.ready < k:
            self._broadcast(k)

    def release(self, k: int, stream_set: int = 0) -> None:
        if self.collective and self.gpu:
            b = self.slot(k)
            ev = self.free_pool[b][self.uses[b] & 1]
            self.uses[b] += 1
            self.engine.record_event(ev, stream_set)
            self.free[b] = ev


def gather_results(local: np.ndarray, frames: Sequence[int], n_frames: int, dst: int = 0, device=None):
    """Gather per-frame result arrays (all of equal length) to rank `dst`.
    `local[k]` are the results of frame `frames[k]`.  Returns a list indexed by frame on `dst`,
    None elsewhere."""
    w = world()
    if w == 1:
        out = [None] * n_frames
        for f, r in zip(frames, local):
            out[f] = r
        return out
    per_frame = local[0].view(np.uint8).size if len(local) else 0
    t = torch.tensor([per_frame], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_frame = int(t.item())
    max_frames = (n_frames + w - 1) // w
    buf = torch.zeros(max_frames * per_frame, dtype=torch.uint8, device=device)
    for k, r in enumerate(local):
        buf[k * per_frame:(k + 1) * per_frame] = torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).reshape(-1)).to(buf.device)
    parts = [torch.zeros_like(buf) for _ in range(w)] if rank() == dst else None
    dist.gather(buf, parts, dst=dst)
    if rank() != dst:
        return None
    out = [None] * n_frames
    dt = local[0].dtype if len(local) else np.uint8
    for r_ in range(w):
        for k, f in enumerate(shard_frames(n_frames, r_, w)):
            out[f] = parts[r_][k * per_frame:(k + 1) * per_frame].cpu().numpy().view(dt)
    return out
