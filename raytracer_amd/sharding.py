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
    """

    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = rank, world, group
        self._pending = []      # (work, recv_list or None)
        self._pool = []         # receive lists allocated ahead of time (rank 0)

    def prealloc(self, n_submits: int, like: torch.Tensor):
        """Allocates rank 0's receive buffers for the next n_submits calls up front, so that no
        device allocation happens between submit() calls."""
        if self.rank == 0 and self.world > 1:
            self._pool = [[torch.empty_like(like) for _ in range(self.world)] for _ in range(n_submits)]

    def submit(self, batch: torch.Tensor):
        if self.world == 1:
            self._pending.append((None, [batch]))
            return
        recv = None
        if self.rank == 0:
            recv = self._pool.pop() if self._pool else [torch.empty_like(batch) for _ in range(self.world)]
        work = dist.gather(batch, gather_list=recv, dst=0, group=self.group, async_op=True)
        self._pending.append((work, recv))

    def finish(self) -> Optional[List[List[torch.Tensor]]]:
        out = []
        for work, recv in self._pending:
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
    """
    mine = frames_for_rank(n_frames, rank, world)
    per_rank = (n_frames + world - 1) // world            # the longest share
    n_batches = (per_rank + batch - 1) // batch
    g = FrameGather(rank, world, group)
    for b in range(n_batches):
        buf = torch.zeros((batch, height, width, 3), dtype=torch.uint8, device=device)
        for k in range(batch):
            i = b * batch + k
            if i < len(mine):
                render_one(mine[i], buf[k])
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
