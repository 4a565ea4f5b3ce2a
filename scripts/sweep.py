#!/usr/bin/env python3
"""Times kernel variants against each other in ONE process, interleaved over several rounds
(kernel time from HIP events on the launch stream), and checks that every variant produces the
same bytes as the first one.

    python scripts/sweep.py --configs 0:256,0:64,1:1024 --rounds 5 --frames 0,45 --out gpurun_out/sweep.json
"""
import argparse
import json
import os
import sys

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
    p.add_argument("--reps", type=int, default=8)
    p.add_argument("--out", default="gpurun_out/sweep.json")
    a = p.parse_args()
    from raytracer_amd import demo
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    ctx.set_option("timing", 0)
    stream = torch.cuda.Stream(device=dev)
    frames = [int(f) for f in a.frames.split(",")]
    configs = []
    for c in a.configs.split(","):
        parts = [int(x) for x in c.split(":")]
        configs.append((parts[0], parts[1], parts[2] if len(parts) > 2 else 1))
    W, H = a.width, a.height
    ref = {}
    buf = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    times = {c: {f: [] for f in frames} for c in configs}
    same = {c: True for c in configs}
    for r in range(a.rounds + 1):                         # round 0 = warm-up + parity check
        for c in configs:
            try:
                ctx.set_option("variant", c[0])
                ctx.set_option("block", c[1])
                ctx.set_option("coop", c[2])
            except Exception as e:                        # unknown variant/block: skip
                same[c] = f"unsupported: {e}"
                continue
            for f in frames:
                if r == 0:
                    scene.render(f, W, H, buf, stream=stream.cuda_stream)
                    stream.synchronize()
                    if f not in ref:
                        ref[f] = buf.clone()
                    elif not torch.equal(ref[f], buf):
                        same[c] = False
                else:
                    # a.reps back-to-back launches between two events on the launch stream
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(a.reps):
                        scene.render(f, W, H, buf, stream=stream.cuda_stream)
                    e1.record(stream)
                    e1.synchronize()
                    times[c][f].append(e0.elapsed_time(e1) / a.reps)
    out = []
    for c in configs:
        row = {"variant": c[0], "block": c[1], "coop": c[2], "same_bytes_as_first": same[c]}
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
