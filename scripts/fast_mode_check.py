#!/usr/bin/env python3
"""The fast build of the kernels (context option "fast") against the exact build: share of camera rays with the same
hit id, share of pixels whose 8-bit colours differ by at most 1, and the time of the batched launches of bench.py."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import bench
    from raytracer_amd import demo
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    W, H = 1920, 1080
    dev = "cuda:0"
    rows = []
    for f in (0, 24, 45, 72, 95):
        planes = {}
        for fast in (0, 1):
            ctx.set_option("fast", fast)
            rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
            hid = torch.empty((H, W), dtype=torch.int32, device=dev)
            scene.render(f, W, H, rgb, hit_id=hid)
            ctx.sync()
            planes[fast] = (rgb.cpu().numpy().astype(np.int16), hid.cpu().numpy())
        same_id = float((planes[0][1] == planes[1][1]).mean())
        diff = np.abs(planes[0][0] - planes[1][0]).max(axis=2)
        rows.append({"frame": f, "same_hit_id": round(same_id, 7), "ids_differing": int((planes[0][1] != planes[1][1]).sum()),
                     "rgb_within_1": round(float((diff <= 1).mean()), 7), "rgb_equal": round(float((diff == 0).mean()), 5),
                     "max_rgb_diff": int(diff.max())})
    seq = bench.frame_sequence(list(range(demo.N_FRAMES)), 96)
    buf = torch.empty((8, H, W, 3), dtype=torch.uint8, device=dev)
    ms = {}
    for fast in (0, 1, 0, 1):
        ctx.set_option("fast", fast)
        t = []
        for b in range(0, 96, 8):
            scene.render_frames(seq[b:b + 8], W, H, buf)
            t.append(ctx.last_kernel_ms())
        ms[fast] = round(float(np.mean(t)), 4)
    print(json.dumps({"frames": rows, "batch_ms_exact": ms[0], "batch_ms_fast": ms[1], "speedup": round(ms[0] / ms[1], 3)}))
    scene.close()


if __name__ == "__main__":
    main()
