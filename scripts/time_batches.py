#!/usr/bin/env python3
"""Kernel time of the headline launches: the 96 animation frames at 1920x1080 in launches of 8 (bench.py's frame order), HIP events
around each kernel, several passes; prints the mean launch time of the best and of the median pass.  For comparing builds of the
library on one machine (scripts/ab_kernels.sh)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    import bench
    from raytracer_amd import demo
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    for opt in sys.argv[3:]:
        k, v = opt.split("=")
        ctx.set_option(k, int(v))
    W, H = 1920, 1080
    seq = bench.frame_sequence(list(range(96)), 96)
    batches = [seq[i:i + 8] for i in range(0, 96, 8)]
    buf = torch.empty((8, H, W, 3), dtype=torch.uint8, device="cuda:0")
    ctx.set_option("timing", 0)
    for _ in range(3):
        for b in batches:
            scene.render_frames(b, W, H, buf)
    ctx.sync()
    ctx.set_option("timing", 1)
    per_pass = []
    for _ in range(passes):
        ms = []
        for b in batches:
            scene.render_frames(b, W, H, buf)
            ms.append(ctx.last_kernel_ms())
        per_pass.append(float(np.mean(ms)))
    print(json.dumps({"tag": tag, "best_ms": round(min(per_pass), 4), "median_ms": round(float(np.median(per_pass)), 4),
                      "sha": __import__("hashlib").sha256(buf.cpu().numpy().tobytes()).hexdigest()[:16]}), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
