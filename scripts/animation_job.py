#!/usr/bin/env python3
"""The reference driver's whole job (raytracer.c:8-75: 96 frames of the drone / treasure / ground animation, then the
animated WebP) dealt over N GPUs: frame f is rendered by rank f mod N (batches of frames through rt_render_frames), every
finished batch is gathered to rank 0 (torch.distributed: RCCL by default) while the next one renders, and rank 0 upscales
the gathered frames on its GPU and hands them to the animation writer while later batches are still on their way
(scene.c:162-225).  Strong scaling: the job is fixed, the GPUs share it.

    python scripts/animation_job.py --gpus 8                       # starts its own 8 ranks
    python scripts/animation_job.py --width 1920 --height 1080 --scale 1.0 --gpus 8

Rank 0 prints one JSON line: seconds from the first launch to the last gathered frame and to the closed file, frames per
second, and the SHA-256 over all gathered frames in frame order -- at the driver's own size (720x540) it is compared with the
hash of the compiled reference's 96 frames (tests/golden/animation_hashes.json)."""
import argparse
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def parse(argv):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--width", type=int, default=720)
    p.add_argument("--height", type=int, default=540)
    p.add_argument("--scale", type=float, default=0.9, help="create_scene's scale_factor: the file is size / scale")
    p.add_argument("--batch", type=int, default=0, help="frames per launch and gather message; 0: up to 16, but at least two messages per rank so that one travels while the next renders")
    p.add_argument("--out", default="", help="animation file (default: none written, frames are hashed only)")
    p.add_argument("--hash", action="store_true", help="hash the gathered frames at sizes without a stored reference hash too")
    p.add_argument("--backend", default="nccl")
    p.add_argument("--one-device", action="store_true", help="rehearsal: every rank on GPU 0 (with --backend gloo)")
    p.add_argument("--master-port", type=int, default=0)
    p.add_argument("--rehearse-cpu", action="store_true",
                   help="no GPU and no rendering: frames are byte patterns in host memory, gathered over gloo and checked frame by frame "
                        "on rank 0 -- the ranks, the frame dealing, the padding and the gathers of an N-GPU run, on a machine without GPUs")
    p.add_argument("--collective-at-one-rank", action="store_true",
                   help="started under torch.distributed.run with one rank: create the process group and gather through it all the same")
    return p.parse_args(argv)


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        import bench
        return bench.spawn_ranks(a, argv, script=__file__, marker='"job"')
    import torch
    import torch.distributed as dist
    from raytracer_amd import animation, demo, sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = a.rehearse_cpu
    if rehearse:
        a.backend = "gloo"
    elif not torch.cuda.is_available():
        sys.exit("animation_job.py: no GPU visible; the rendering core has no CPU path")
    if rehearse:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    grouped = world > 1 or (a.collective_at_one_rank and "WORLD_SIZE" in os.environ)
    if grouped:
        dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
    W, H, NF = a.width, a.height, demo.N_FRAMES
    if rehearse:
        import bench
        scene, ctx = bench.PatternScene(), None
    else:
        scene = demo.DemoScene(local)
        ctx = scene.ctx
        ctx.set_option("timing", 0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)                  # kernels, gathers and the upscale are ordered by this stream
    sptr = None if rehearse else stream.cuda_stream

    def device_sync():
        if not rehearse:
            torch.cuda.synchronize(dev)
    mine = sharding.frames_for_rank(NF, rank, world)
    longest = (NF + world - 1) // world
    B = a.batch if a.batch > 0 else max(1, min(16, (longest + 1) // 2))
    n_batches = (longest + B - 1) // B
    bufs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(min(n_batches, 4))]
    gold = None
    try:
        with open(os.path.join(REPO, "tests", "golden", "animation_hashes.json")) as f:
            g = json.load(f)
        if (g["width"], g["height"]) == (W, H):
            gold = g["all_frames_sha256"]
    except OSError:
        pass
    hashing = rank == 0 and (a.hash or gold is not None)      # a host-side pass over every frame: only where it checks something
    sha = hashlib.sha256()
    seen = [0]
    pattern_ok = [0]
    sink = None
    if rank == 0 and a.out and not rehearse:
        sink = animation.AnimationSink(ctx, a.out, W, H, a.scale, NF, 4000)

    def consume(_seq, per_rank):
        frames = torch.stack([t.to(dev) for t in per_rank], dim=1).reshape(-1, H, W, 3)       # frame = (batch * B + i) * world + r
        n = min(int(frames.shape[0]), NF - seen[0])
        if n > 0:
            if rehearse:                               # frame f carries the byte PatternScene gives it: every pixel of every frame, in order
                for i in range(n):
                    f = seen[0] + i
                    pattern_ok[0] += int(bool((frames[i] == (37 * f + 11) % 251).all()))
            if hashing:
                sha.update(frames[:n].cpu().numpy().tobytes())
            if sink is not None:
                sink.add(frames[:n])
            seen[0] += n

    # one untimed launch: code objects and clocks
    scene.render_frames([mine[0] if mine else 0], W, H, bufs[0][:1], stream=sptr)
    device_sync()
    if grouped:
        dist.barrier()
    gather = sharding.FrameGather(rank, world, ring=3, consume=consume, collective_at_one_rank=grouped)
    gather.prealloc(n_batches, bufs[0])
    t0 = time.perf_counter()
    for b in range(n_batches):
        buf = bufs[b % len(bufs)]
        frames = [mine[b * B + k] if b * B + k < len(mine) else (mine[-1] if mine else 0) for k in range(B)]    # padding: dropped by rank 0
        scene.render_frames(frames, W, H, buf, stream=sptr)
        gather.submit(buf)
    gather.finish()
    device_sync()
    t_frames = time.perf_counter() - t0
    if sink is not None:
        sink.close()
    t_all = time.perf_counter() - t0
    if grouped:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"job": "the reference driver's animation, frame f on rank f mod N, framebuffers gathered to rank 0",
                          "frames": seen[0], "width": W, "height": H, "n_gpus": world, "backend": dist.get_backend() if grouped else None,
                          "frames_per_launch": B, "seconds_until_all_frames_on_rank0": round(t_frames, 4),
                          "seconds_until_file_closed": round(t_all, 4) if sink is not None else None,
                          "frames_per_s": round(seen[0] / t_frames, 1), "all_frames_sha256": sha.hexdigest() if hashing else None,
                          "equals_compiled_reference": (sha.hexdigest() == gold) if gold else None,
                          "file": a.out or None,
                          **({"rehearsal": "cpu: byte patterns over gloo, no GPU -- not a measurement", "frames_verified_in_order": pattern_ok[0]}
                             if rehearse else {})}), flush=True)
    scene.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
