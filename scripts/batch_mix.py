#!/usr/bin/env python3
"""rt_render_frames on batches of neighbouring frames against batches of frames from all over the animation
(what a rank of an 8-GPU run holds: frames r, r+8, ...), with one starting row per batch or per frame."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from raytracer_amd import demo
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    W, H = 1920, 1080
    buf = torch.empty((8, H, W, 3), dtype=torch.uint8, device="cuda:0")
    sets = {"neighbours": [list(range(b, b + 8)) for b in range(0, 96, 8)],
            "stride7": [[(7 * k) % 96 for k in range(b, b + 8)] for b in range(0, 96, 8)],
            "stride8": [[(r + 8 * k) % 96 for k in range(8)] for r in range(8)]}
    out = {}
    for rows in (0, 1):
        ctx.set_option("batch_rows", rows)
        for name, batches in sets.items():
            for _ in range(2):
                ms = []
                for fr in batches:
                    scene.render_frames(fr, W, H, buf)
                    ms.append(ctx.last_kernel_ms())
            out[f"{name}_rows{rows}"] = round(float(np.mean(ms)), 4)
    print(json.dumps(out))
    scene.close()


if __name__ == "__main__":
    main()
