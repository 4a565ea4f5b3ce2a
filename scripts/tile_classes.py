#!/usr/bin/env python3
"""Time of every 8x8 tile of a 1920x1080 frame (rt_set_trace_buffer), summed by what the tile's camera rays hit (sky, ground
only, drone, treasure ...).  Tile times are residency, not instruction counts: a wave waiting for memory counts too.
Run on a GPU: python scripts/tile_classes.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import demo
scene = demo.DemoScene(0); ctx = scene.ctx
W, H = 1920, 1080
tx, ty = (W + 7) // 8, (H + 7) // 8
buf = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
hid = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
ctx.set_option("variant", 0); ctx.set_option("block", 64)
for frame in (0, 45):
    clk = torch.zeros((tx * ty, 2), dtype=torch.int64, device="cuda:0")
    for rep in range(3):
        ctx.set_trace_buffer(clk if rep == 2 else None)
        scene.render(frame, W, H, buf, hit_id=hid); ctx.sync()
    ctx.set_trace_buffer(None)
    t = clk.cpu().numpy().astype(np.float64) / 100.0
    dur = (t[:, 1] - t[:, 0])
    ids = hid.cpu().numpy()
    mesh = np.where(ids >= 0, ids >> 24, 3)
    # classify tiles by meshes present among hit ids
    cls = np.zeros(tx * ty, int)
    for j in range(ty):
        for i in range(tx):
            m = mesh[j * 8:(j + 1) * 8, i * 8:(i + 1) * 8]
            c = 0
            if (m == 0).any(): c |= 1
            if (m == 1).any(): c |= 2
            if (m == 2).any(): c |= 4
            cls[j * tx + i] = c
    tot = dur.sum()
    out = {"frame": frame, "sum_tile_us": round(tot, 0)}
    for c, name in ((0, "sky"), (4, "ground only"), (1, "drone"), (5, "drone+ground"), (2, "treasure"), (6, "treasure+ground"), (3, "d+t"), (7, "all")):
        sel = cls == c
        if sel.any():
            out[name] = {"tiles": int(sel.sum()), "share_of_time": round(float(dur[sel].sum() / tot), 3), "mean_us": round(float(dur[sel].mean()), 1), "max_us": round(float(dur[sel].max()), 1)}
    print(json.dumps(out))
scene.close()
