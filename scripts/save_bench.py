#!/usr/bin/env python3
"""Times the reference driver's whole job with the frames consumed as they are rendered: 96 frames at
720x540, upscaled on the GPU to 800x600 and written as an animated WebP by the host library's
encoder threads (the reference: save_scene after the last render_scene, one thread).  One JSON line.

    python scripts/save_bench.py --quality 75 --threads 16
"""
import argparse
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=800)
    p.add_argument("--height", type=int, default=600)
    p.add_argument("--scale", type=float, default=0.9)
    p.add_argument("--frames", type=int, default=96)
    p.add_argument("--duration", type=int, default=4000)
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--quality", type=float, default=75.0)
    p.add_argument("--threads", type=int, default=16)
    p.add_argument("--out", default="/tmp/save_bench.webp")
    a = p.parse_args()
    from raytracer_amd import animation, demo, sharding
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = demo.DemoScene(0)
    w, h = int(a.width * a.scale), int(a.height * a.scale)
    bufs = [torch.empty((a.batch, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
    # an explicit stream, made torch's current one: renders, torch's own kernels and the upscale are then
    # ordered on it (a null handle would mean the library's private stream, which torch does not wait for)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    scene.render_frames(list(range(a.batch)), w, h, bufs[0], stream=stream)             # warm-up
    torch.cuda.synchronize(dev)
    rows = {}
    for label, threads in (("pipelined", a.threads), ("one_encoder_thread", 1)):
        t0 = time.perf_counter()
        with animation.AnimationSink(scene.ctx, a.out, w, h, a.scale, a.frames, a.duration, quality=a.quality, threads=threads) as sink:
            gather = sharding.FrameGather(0, 1, ring=2, consume=sink.consume_gathered(1))
            nb = (a.frames + a.batch - 1) // a.batch
            for b in range(nb):
                fr = [min(b * a.batch + k, a.frames - 1) for k in range(a.batch)]
                scene.render_frames(fr, w, h, bufs[b % 3], stream=stream)
                gather.submit(bufs[b % 3])
            gather.finish()
            t_fed = time.perf_counter()
        t1 = time.perf_counter()
        rows[label] = {"threads": threads, "total_s": round(t1 - t0, 3), "until_last_frame_handed_over_s": round(t_fed - t0, 3),
                       "file_bytes": os.path.getsize(a.out), "file_sha1": hashlib.sha1(open(a.out, "rb").read()).hexdigest()}
    print(json.dumps({"workload": f"{a.frames} frames {w}x{h} -> upscaled by 1/{a.scale}, quality {a.quality}", **rows}), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
