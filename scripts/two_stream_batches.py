#!/usr/bin/env python3
"""Do the tails of consecutive rt_render_frames launches cost anything?  The 96 animation frames at 1920x1080 in launches of 16
(bench.py's frame order), all on one stream against alternating between two (a launch's last tiles then run beside the next launch's
first ones); wall time of the whole loop, several passes, best and median."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import bench
    from raytracer_amd import demo
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    ctx.set_option("timing", 0)
    W, H, B = 1920, 1080, 16
    seq = bench.frame_sequence(list(range(96)), 96)
    batches = [seq[i:i + B] for i in range(0, 96, B)]
    bufs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    out = {}
    for n_streams in (1, 2, 3, 1, 2, 3):
        use = streams[:n_streams]
        for _ in range(3):
            for k, b in enumerate(batches):
                scene.render_frames(b, W, H, bufs[k % 4], stream=use[k % n_streams].cuda_stream)
        torch.cuda.synchronize()
        times = []
        for _ in range(9):
            t0 = time.perf_counter()
            for k, b in enumerate(batches):
                scene.render_frames(b, W, H, bufs[k % 4], stream=use[k % n_streams].cuda_stream)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        out.setdefault(f"{n_streams}_streams_ms_per_96_frames", []).append([round(min(times), 3), round(float(np.median(times)), 3)])
    print(json.dumps(out), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
