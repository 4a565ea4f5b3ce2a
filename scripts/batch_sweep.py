#!/usr/bin/env python3
"""Times rt_render_frames for several batch sizes against single-frame launches, and checks that
a batch writes the same bytes as the single-frame path.

    python scripts/batch_sweep.py --batches 1,2,4,8,16 --blocks 64,256 --streams 1,2 --out gpurun_out/batch.json
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batches", default="1,2,4,8,16")
    p.add_argument("--blocks", default="256")
    p.add_argument("--streams", default="1")
    p.add_argument("--start", type=int, default=0, help="first animation frame; frames cycle through the 96 of the animation")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--total", type=int, default=96, help="frames rendered per measurement")
    p.add_argument("--rounds", type=int, default=3)
    p.add_argument("--swizzle", default="1", help="comma list of 0/1: XCD-aware tile order")
    p.add_argument("--chunks", default="0", help="comma list: consecutive tiles of one frame before the next frame's")
    p.add_argument("--out", default="gpurun_out/batch.json")
    a = p.parse_args()
    from raytracer_amd import demo
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    ctx.set_option("timing", 0)
    ctx.set_option("variant", 0)
    W, H = a.width, a.height
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    maxb = max(int(b) for b in a.batches.split(","))
    bufs = [torch.empty((maxb, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
    one = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)

    # parity: a batch of frames start..start+maxb-1 against single launches
    frames = [(a.start + i) % demo.N_FRAMES for i in range(maxb)]
    ok = True
    for block in [int(b) for b in a.blocks.split(",")]:
        ctx.set_option("block", block)
        bufs[0].zero_()
        scene.render_frames(frames, W, H, bufs[0], stream=streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        for i, f in enumerate(frames):
            scene.render(f, W, H, one, stream=streams[0].cuda_stream)
            torch.cuda.synchronize(dev)
            if not torch.equal(one, bufs[0][i]):
                ok = False
                print(f"MISMATCH block {block} frame {f}", flush=True)
    print(json.dumps({"batch_equals_single": ok}), flush=True)

    rows = []
    for block, chunk, swz in [(int(b), int(q), int(z)) for b in a.blocks.split(",") for q in a.chunks.split(",") for z in a.swizzle.split(",")]:
        ctx.set_option("block", block)
        ctx.set_option("batch_chunk", chunk)
        ctx.set_option("xcd_swizzle", swz)
        for ns in [int(s) for s in a.streams.split(",")]:
            for nb in [int(b) for b in a.batches.split(",")]:
                launches = max(a.total // nb, 1)
                ts = []
                for r in range(a.rounds + 1):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    f = a.start
                    for k in range(launches):
                        fr = [(f + i) % demo.N_FRAMES for i in range(nb)]
                        f += nb
                        scene.render_frames(fr, W, H, bufs[k % ns], stream=streams[k % ns].cuda_stream)
                    torch.cuda.synchronize(dev)
                    if r:
                        ts.append((time.perf_counter() - t0) * 1e3 / (launches * nb))
                row = {"block": block, "chunk": chunk, "swizzle": swz, "streams": ns, "batch": nb, "ms_per_frame_median": round(float(np.median(ts)), 4),
                       "ms_per_frame_min": round(float(np.min(ts)), 4)}
                rows.append(row)
                print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"batch_equals_single": ok, "rows": rows}, open(a.out, "w"), indent=1)
    scene.close()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
