#!/usr/bin/env python3
"""24 instanced frames on one stream and alternating between two streams: if the frame were bound by its slowest tiles, two
frames in flight would overlap.  Argument: option instanced_wave.  Run on a GPU: python scripts/instanced_overlap.py 0"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import capi, demo
scene = demo.DemoScene(0); ctx = scene.ctx
ctx.set_option("instanced_wave", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx.set_option("timing", 0)
rng = np.random.default_rng(3)
ids, tfs = [0, 2], [scene.transforms(0)[0], capi.Transform()]
for gx in range(56):
    for gz in range(56):
        ids.append(1)
        tfs.append(capi.Transform(capi.Vec3((gx - 27.5) * 1.05, 0.45 + 0.1 * rng.uniform(), (gz - 27.5) * 1.05), capi.Vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
W, H = 1920, 1080
inst = capi.Context.make_instances(ids, tfs)
out = {}
for ns in (1, 2):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    bufs = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(ns)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(24):
            ctx.render_instanced(scene.camera, scene.light, inst, W, H, bufs[k % ns], stream=streams[k % ns].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 24 * 1e3
    out[f"streams{ns}_ms_per_frame"] = round(dt, 3)
print(json.dumps(out))
scene.close()
