"""Rank 0's consumer of gathered frames: GPU upscale, then the animation file.

The reference keeps all frames in memory and encodes them one after the other once rendering is done
(save_scene, scene.c:162-225).  Here frames are consumed as their gather completes
(sharding.FrameGather's `consume` hook): a batch is upscaled on the GPU to the encoder's size
(scene.c:164-165, 188-202), copied to the host once, and handed to the host library's animation
writer, whose threads encode while later frames are still being rendered and gathered.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import capi, host_api


def frame_times(frame_count: int, duration_ms: int) -> List[int]:
    """How long each frame is shown: the reference stamps frame f at f*(duration_ms/frame_count) and ends
    the animation at duration_ms (scene.c:204, 209)."""
    if frame_count <= 0:
        return []
    step = duration_ms // frame_count
    return [step] * (frame_count - 1) + [duration_ms - (frame_count - 1) * step]


class AnimationSink:
    """Writes an animation from batches of device frames in frame order.

    ctx            capi.Context on the device the frames live on
    width/height   rendered size; scale_factor as in create_scene (output = size / scale_factor, rounded)
    """

    def __init__(self, ctx: "capi.Context", filename: str, width: int, height: int, scale_factor: float,
                 frame_count: int, duration_ms: int, quality: float = 75.0, threads: int = 0):
        self.ctx, self.width, self.height = ctx, int(width), int(height)
        self.scaled = capi.scaled_size(width, height, scale_factor)
        self.times = frame_times(frame_count, duration_ms)
        self.frame_count = frame_count
        self.written = 0
        self._h = C.c_void_p()
        rc = host_api.lib().rt_host_animation_open(filename.encode(), self.scaled[0], self.scaled[1], float(quality), int(threads),
                                                   C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"cannot write {filename}: {capi.status_string(rc)}")
        self._argb = None

    def add(self, frames: torch.Tensor):
        """frames: [n, height, width, 3] uint8 on the GPU, the next n frames of the animation.  Frames beyond
        frame_count (padding of the last batch) are dropped."""
        n = min(int(frames.shape[0]), self.frame_count - self.written)
        if n <= 0:
            return
        if tuple(frames.shape[1:]) != (self.height, self.width, 3) or frames.dtype != torch.uint8 or not frames.is_cuda:
            raise ValueError("frames must be a CUDA uint8 tensor [n, height, width, 3]")
        frames = frames.contiguous()
        sw, sh = self.scaled
        if self._argb is None or self._argb.shape[0] < n:
            self._argb = torch.empty((n, sh, sw), dtype=torch.int32, device=frames.device)
        # The frames are complete once torch's current stream has drained (they may come from a collective
        # on it).  A stream handle of 0 means "the context's own stream" to the library, which torch's copies
        # do not wait for: so wait for the upscale explicitly before copying its output.
        cur = torch.cuda.current_stream(frames.device)
        cur.synchronize()
        self.ctx.upscale_frames(frames, n, self.width, self.height, sw, sh, self._argb, stream=cur.cuda_stream)
        self.ctx.sync(cur.cuda_stream)
        host = self._argb[:n].cpu()
        L = host_api.lib()
        for k in range(n):
            rc = L.rt_host_animation_add(self._h, host[k].data_ptr(), 1, self.times[self.written])
            if rc != 0:
                raise RuntimeError(f"animation writer: {capi.status_string(rc)}")
            self.written += 1

    def consume_gathered(self, world: int):
        """A `consume` callback for FrameGather when frame f is rendered by rank f mod world: the per-rank
        batches of one gather interleave to one run of consecutive frames."""
        def consume(_seq, per_rank):
            stacked = torch.stack(list(per_rank), dim=1)              # [B, world, H, W, 3]: frame = (seq*B + i)*world + r
            self.add(stacked.reshape(-1, self.height, self.width, 3))
        return consume

    def close(self):
        if self._h:
            rc = host_api.lib().rt_host_animation_close(self._h)
            self._h = C.c_void_p()
            if rc != 0:
                raise RuntimeError(f"animation writer: {capi.status_string(rc)}")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
