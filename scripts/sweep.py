#!/usr/bin/env python3
"""Times kernel variants against each other in ONE process, interleaved over several rounds, and
checks that every variant produces the same bytes as the first one.

A config is variant:block[:budget[:streams]].  With streams > 1 consecutive frames are launched
round-robin on that many HIP streams (frames are independent, so they may overlap on the GPU);
the time reported is wall time of `reps` frames divided by `reps`, bracketed by device syncs.

    python scripts/sweep.py --configs 0:256,0:256:48:4,4:256:32 --rounds 5 --frames 0,45 --out gpurun_out/sweep.json
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
    p.add_argument("--configs", default="0:256")
    p.add_argument("--rounds", type=int, default=5)
    p.add_argument("--frames", default="0,45")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--reps", type=int, default=16)
    p.add_argument("--out", default="gpurun_out/sweep.json")
    a = p.parse_args()
    from raytracer_amd import demo
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    ctx.set_option("timing", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    frames = [int(f) for f in a.frames.split(",")]
    configs = []
    for c in a.configs.split(","):
        parts = [int(x) for x in c.split(":")]
        configs.append((parts[0], parts[1], parts[2] if len(parts) > 2 else 48, parts[3] if len(parts) > 3 else 1,
                        parts[4] if len(parts) > 4 else 32))
    W, H = a.width, a.height
    ref = {}
    bufs = [torch.empty((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(8)]
    times = {c: {f: [] for f in frames} for c in configs}
    same = {c: True for c in configs}
    for r in range(a.rounds + 1):                         # first pass = warm-up + parity check
        for c in configs:
            try:
                ctx.set_option("variant", c[0])
                ctx.set_option("block", c[1])
                ctx.set_option("budget", c[2])
                ctx.set_option("resident_budget", c[4] * 1024)
            except Exception as e:                        # unknown variant/block: skip
                same[c] = f"unsupported: {e}"
                continue
            ns = c[3]
            for f in frames:
                if r == 0:
                    scene.render(f, W, H, bufs[0], stream=streams[0].cuda_stream)
                    torch.cuda.synchronize(dev)
                    if f not in ref:
                        ref[f] = bufs[0].clone()
                    elif not torch.equal(ref[f], bufs[0]):
                        same[c] = False
                else:
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for k in range(a.reps):
                        scene.render(f, W, H, bufs[k % ns], stream=streams[k % ns].cuda_stream)
                    torch.cuda.synchronize(dev)
                    times[c][f].append((time.perf_counter() - t0) * 1e3 / a.reps)
    out = []
    for c in configs:
        row = {"variant": c[0], "block": c[1], "budget": c[2], "streams": c[3], "resident_kb": c[4], "same_bytes_as_first": same[c]}
        for f in frames:
            t = times[c][f]
            if t:
                row[f"f{f}_ms_median"] = round(float(np.median(t)), 4)
                row[f"f{f}_ms_min"] = round(float(np.min(t)), 4)
        out.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    scene.close()


if __name__ == "__main__":
    main()
