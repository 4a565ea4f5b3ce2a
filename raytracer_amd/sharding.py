"""Frame sharding over ranks and the framebuffer gather to rank 0.

Animation frames are independent (each render reads only its own per-frame placements and writes
its own buffer, reference raytracer.c:40-57), so the animation is dealt round-robin: frame f is
rendered by rank f mod N.  The only exchange step is the gather of finished RGB8 framebuffers to
rank 0, where the reference's encoder would consume them (scene.c:186-206).  It runs as
torch.distributed gathers (RCCL over xGMI on GPUs, gloo on CPU tensors) of whole batches of frames,
issued asynchronously so that a batch travels while the next one renders.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Frames of an n_frames animation owned by `rank`: rank, rank+world, ..."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, max(n_frames, 0), world))


def owner_of(frame: int, world: int) -> int:
    return frame % world


class FrameGather:
    """Collects per-rank batches of frames on rank 0.

    Every rank calls submit() the same number of times with a [B, H, W, 3] uint8 tensor (the same
    shape on every rank); rank 0 receives all ranks' batches.  finish() waits for the transfers and,
    on rank 0, returns for each submit call the list of per-rank tensors.

    With `ring` set, at most that many gathers are kept in flight: submit() first waits for the
    oldest one, hands its frames to `consume` (rank 0; default: drop them) and reuses its receive
    buffers.  A caller that renders into a ring of `ring` send buffers may therefore overwrite a
    buffer as soon as the submit() that follows it in the ring has returned.
    """

    def __init__(self, rank: int, world: int, group=None, ring: Optional[int] = None,
                 consume: Optional[Callable[[int, List[torch.Tensor]], None]] = None, collective_at_one_rank: bool = False):
        self.rank, self.world, self.group = rank, world, group
        # a single rank needs no transfer; with this set it issues the collective all the same (a one-rank process group must
        # exist): the way to run the backend's gather on a machine with one GPU
        self.solo = world == 1 and not collective_at_one_rank
        self.ring, self.consume = ring, consume
        self._pending = []      # (sequence number, work, recv_list or None)
        self._pool = []         # receive lists ready for reuse (rank 0)
        self._seq = 0

    def prealloc(self, n_submits: int, like: torch.Tensor):
        """Allocates rank 0's receive buffers for the next n_submits calls (or for the whole ring) up
        front, so that no device allocation happens between submit() calls."""
        if self.rank == 0 and not self.solo:
            n = min(n_submits, self.ring) if self.ring else n_submits
            self._pool = [[torch.empty_like(like) for _ in range(self.world)] for _ in range(n)]

    def _retire_oldest(self):
        seq, work, recv = self._pending.pop(0)
        if work is not None:
            work.wait()
        if self.rank == 0 and recv is not None:
            if self.consume is not None:
                self.consume(seq, recv)
            if not self.solo:
                self._pool.append(recv)

    def submit(self, batch: torch.Tensor):
        if self.ring and len(self._pending) >= self.ring:
            self._retire_oldest()
        seq = self._seq
        self._seq += 1
        if self.solo:
            self._pending.append((seq, None, [batch]))
            return
        if batch.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no gather for device tensors: go through host memory (rehearsals of the
            # multi-rank flow on machines without RCCL peers; the GPU path proper is the branch below)
            batch = batch.cpu()
            self._pool = [p for p in self._pool if not p[0].is_cuda]
        recv = None
        if self.rank == 0:
            recv = self._pool.pop() if self._pool else [torch.empty_like(batch) for _ in range(self.world)]
        work = dist.gather(batch, gather_list=recv, dst=0, group=self.group, async_op=True)
        self._pending.append((seq, work, recv))

    def finish(self) -> Optional[List[List[torch.Tensor]]]:
        """Waits for everything in flight.  Without a ring: returns rank 0's received batches in submit
        order.  With a ring: the remaining batches go to `consume` and nothing is returned."""
        if self.ring:
            while self._pending:
                self._retire_oldest()
            return None
        out = []
        for _, work, recv in self._pending:
            if work is not None:
                work.wait()
            out.append(recv)
        self._pending = []
        return out if self.rank == 0 else None


def render_sharded(n_frames: int, rank: int, world: int, height: int, width: int,
                   render_one: Callable[[int, torch.Tensor], None], device, batch: int = 4, group=None):
    """Renders this rank's share of an n_frames animation and gathers everything on rank 0.

    render_one(frame_index, out[H,W,3] uint8 on `device`) enqueues one frame.  Ranks with fewer
    frames than others pad their last batch with frames that rank 0 discards, so that every rank
    issues the same collectives.  Returns on rank 0 a dict frame -> tensor [H,W,3]; None elsewhere.

    On a GPU everything here is ordered by torch's CURRENT stream: the padding is cleared on it and the
    gather is queued behind it, so render_one must launch on that stream too (pass
    torch.cuda.current_stream().cuda_stream as the `stream` argument of the rendering call) -- a
    frame rendered on another stream could be overtaken by the gather that sends it.
    """
    mine = frames_for_rank(n_frames, rank, world)
    per_rank = (n_frames + world - 1) // world            # the longest share
    n_batches = (per_rank + batch - 1) // batch
    g = FrameGather(rank, world, group)
    for b in range(n_batches):
        # no fill of the whole buffer: every slot is written exactly once, by the render or by the padding
        buf = torch.empty((batch, height, width, 3), dtype=torch.uint8, device=device)
        for k in range(batch):
            i = b * batch + k
            if i < len(mine):
                render_one(mine[i], buf[k])
            else:
                buf[k].zero_()
        g.submit(buf)
    got = g.finish()
    if rank != 0:
        return None
    frames = {}
    for b, per in enumerate(got):
        for r, t in enumerate(per):
            theirs = frames_for_rank(n_frames, r, world)
            for k in range(batch):
                i = b * batch + k
                if i < len(theirs):
                    frames[theirs[i]] = t[k]
    return frames


# ------------------------------------------------------------------------------------------------
# One frame over several ranks: strips of eight rows dealt round-robin (pixels are independent,
# reference scene.c:58-59).  Rank r renders strips r, r + N, ... with rt_render_frame_strips /
# rt_render_frame_sampled_strips into a plane strip_rows(height, N, r) rows tall; the shares are
# gathered on rank 0 and interleaved back into the frame.
# ------------------------------------------------------------------------------------------------
def strip_rows(height: int, world: int, rank: int) -> int:
    """Rows of a `height`-row frame in strips rank, rank + world, ... (what rt_strip_rows returns)."""
    total = (height + 7) // 8
    if rank >= total:
        return 0
    k = (total - rank + world - 1) // world
    rows = 8 * k
    if rank + (k - 1) * world == total - 1 and height % 8:
        rows -= 8 - height % 8
    return rows


def frame_rows_of_share(height: int, world: int, rank: int) -> torch.Tensor:
    """Frame row of every row of rank's share, in the share's order."""
    n = strip_rows(height, world, rank)
    r = torch.arange(n, dtype=torch.int64)
    return ((r // 8) * world + rank) * 8 + r % 8


def assemble_strips(shares: Sequence[torch.Tensor], height: int) -> torch.Tensor:
    """Puts the ranks' shares ([rows_k or more, width, ...] each; extra rows are padding) back into one frame."""
    world = len(shares)
    first = shares[0]
    out = torch.empty((height,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
    for k, share in enumerate(shares):
        rows = frame_rows_of_share(height, world, k).to(first.device)
        out[rows] = share[:rows.numel()].to(first.device)
    return out


def gather_frame(share: torch.Tensor, height: int, rank: int, world: int, group=None) -> Optional[torch.Tensor]:
    """Collective: every rank passes its share (strip_rows(height, world, rank) rows); rank 0 gets the frame.
    Shares are padded to the longest one so that all ranks send the same shape."""
    if world == 1:
        return assemble_strips([share], height)
    longest = strip_rows(height, world, 0)
    send = share
    if share.shape[0] != longest:
        send = torch.zeros((longest,) + tuple(share.shape[1:]), dtype=share.dtype, device=share.device)
        send[:share.shape[0]] = share
    staged = send.is_cuda and dist.get_backend(group) == "gloo"      # gloo gathers host tensors only
    if staged:
        send = send.cpu()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send.contiguous(), gather_list=recv, dst=0, group=group)
    if rank != 0:
        return None
    return assemble_strips(recv, height)
