#!/usr/bin/env python3
"""Per-tile timing trace of one frame for one or more kernel variants: how long each 8x8 tile took,
when it started, and how many waves were in flight over the frame."""
import argparse, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

def main():
    p = argparse.ArgumentParser()
    p.add_argument("--configs", default="0:256,2:256")
    p.add_argument("--frame", type=int, default=0)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    a = p.parse_args()
    from raytracer_amd import demo
    dev = torch.device("cuda", 0)
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    W, H = a.width, a.height
    tx, ty = (W + 7) // 8, (H + 7) // 8
    buf = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    for c in a.configs.split(","):
        v, b = (int(x) for x in c.split(":"))
        ctx.set_option("variant", v); ctx.set_option("block", b)
        clk = torch.zeros((tx * ty, 2), dtype=torch.int64, device=dev)
        for rep in range(3):
            ctx.set_trace_buffer(clk if rep == 2 else None)
            scene.render(a.frame, W, H, buf)
            ctx.sync()
        ctx.set_trace_buffer(None)
        t = clk.cpu().numpy().astype(np.float64) / 100.0        # microseconds (100 MHz clock)
        t0 = t[:, 0].min()
        start, end = t[:, 0] - t0, t[:, 1] - t0
        dur = end - start
        order = np.argsort(dur)[::-1]
        heavy = [(int(i % tx), int(i // tx), round(float(dur[i]), 1), round(float(start[i]), 1)) for i in order[:8]]
        # waves in flight sampled every 10 us
        span = end.max()
        grid = np.arange(0, span, max(span / 40, 1.0))
        inflight = [int(((start <= g) & (end > g)).sum()) for g in grid]
        out = {"variant": v, "block": b, "kernel_ms": round(scene.ctx.last_kernel_ms(), 4), "span_us": round(float(span), 1),
               "sum_tile_us": round(float(dur.sum()), 1), "mean_us": round(float(dur.mean()), 2),
               "p50": round(float(np.percentile(dur, 50)), 2), "p90": round(float(np.percentile(dur, 90)), 2),
               "p99": round(float(np.percentile(dur, 99)), 2), "max": round(float(dur.max()), 2),
               "heaviest_tiles_x_y_us_start": heavy, "inflight_over_time": inflight,
               "last_start_us": round(float(start.max()), 1)}
        print(json.dumps(out), flush=True)
    scene.close()

if __name__ == "__main__":
    main()
