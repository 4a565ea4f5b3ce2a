#!/usr/bin/env python3
"""How unevenly the instanced stress frame's work falls on the 64 pixels of a tile (= the 64 lanes of a wave).
Renders the frame with the host build of the pixel code (tests/host_emul: the product's rt_instanced.h compiled for the CPU),
records each pixel's box and triangle tests, and prints per 8x8 tile: mean / max of the per-pixel cost.  A wave lasts as long
as its slowest lane, so sum(mean) / sum(max) bounds the lane utilisation of ANY one-pixel-per-lane kernel without refill."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--side", type=int, default=56)
    p.add_argument("--spacing", type=float, default=1.05)
    a = p.parse_args()
    import emul
    from oracle import pyoracle as po
    po.lib()
    meshes = po.default_meshes()
    s = emul.Scene()
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    rng = np.random.default_rng(3)
    ids = [0, 2]
    tfs = [po.animation_transforms(0)[0], po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]
    half = (a.side - 1) / 2.0
    for gx in range(a.side):
        for gz in range(a.side):
            ids.append(1)
            tfs.append(po.Transform(po.vec3((gx - half) * a.spacing, 0.45 + 0.1 * rng.uniform(), (gz - half) * a.spacing),
                                    po.vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
    cam, light = po.default_camera(), po.default_light()
    W, H = a.width, a.height
    plane = np.zeros((H, W, 2), np.uint32)
    s.record_work(plane)
    got = s.render_instanced(cam, light, ids, tfs, W, H)
    s.record_work(None)
    cost = plane[..., 0].astype(np.float64) * 22 + plane[..., 1].astype(np.float64) * 50        # FLOP per box / triangle test
    th, tw = H // 8, W // 8
    tiles = cost[:th * 8, :tw * 8].reshape(th, 8, tw, 8).transpose(0, 2, 1, 3).reshape(th, tw, 64)
    mean, mx = tiles.mean(-1), tiles.max(-1)
    out = {"size": [W, H], "instances": len(ids), "counters": got["counters"],
           "box_tests_per_pixel": float(plane[..., 0].mean()), "tri_tests_per_pixel": float(plane[..., 1].mean()),
           "lane_bound_mean_over_max": float(mean.sum() / mx.sum()),
           "tile_max_over_mean_percentiles": {str(q): float(np.percentile(mx / np.maximum(mean, 1), q)) for q in (10, 50, 90, 99)},
           "pixel_cost_percentiles": {str(q): float(np.percentile(cost, q)) for q in (10, 50, 90, 99, 99.9)},
           "pixel_cost_mean": float(cost.mean())}
    shapes = {}
    for bw, bh in ((8, 8), (16, 4), (4, 16), (32, 2), (64, 1), (2, 32)):
        t = cost[:H // bh * bh, :W // bw * bw].reshape(H // bh, bh, W // bw, bw).transpose(0, 2, 1, 3).reshape(-1, bw * bh)
        shapes[f"{bw}x{bh}"] = float(t.mean(-1).sum() / t.max(-1).sum())
    out["lane_bound_by_tile_shape"] = shapes
    # the same bound if lanes were handed pixels of equal cost (perfect sorting): sort all pixels by cost, cut into 64s
    srt = np.sort(cost.reshape(-1))
    srt = srt[:srt.size // 64 * 64].reshape(-1, 64)
    out["lane_bound_if_sorted_by_cost"] = float(srt.mean(-1).sum() / srt.max(-1).sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
