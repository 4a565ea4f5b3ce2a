#!/usr/bin/env python3
"""Where rt_render_frames spends its time, by leaving parts of the scene out: the whole scene, the ground replaced by two
triangles (its walk costs nothing then), the ground alone, two triangles alone (the per-pixel fixed part), the scene without the drone / without the treasure, each of those two alone.  ms per launch of
8 frames at 1920x1080 over the bench's twelve launches.  Run on a GPU: python scripts/cost_split_frames.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import capi, demo, host_api
import bench
scene = demo.DemoScene(0)
W, H = 1920, 1080
seq = bench.frame_sequence(list(range(96)), 96)
buf = torch.empty((8, H, W, 3), dtype=torch.uint8, device="cuda:0")
def timeit(ctx, cam, light, tf_of):
    ms = []
    for rep in range(2):
        ms = []
        for b in range(0, 96, 8):
            ctx.render_frames(cam, light, [tf_of(f) for f in seq[b:b + 8]], W, H, buf)
            ms.append(ctx.last_kernel_ms())
    return round(float(np.mean(ms)), 4)
out = {"normal": timeit(scene.ctx, scene.camera, scene.light, scene.transforms)}
# same scene with the ground replaced by two triangles over its bounds
g = host_api.mesh_triangles(scene.meshes[2])
p = g[:, 0:9].reshape(-1, 3)
lo, hi = p.min(axis=0), p.max(axis=0)
quad = np.zeros((2, 24), np.float32)
a, b, c, d = [lo[0], 0, lo[2]], [hi[0], 0, lo[2]], [hi[0], 0, hi[2]], [lo[0], 0, hi[2]]
quad[0, 0:9] = a + c + b; quad[1, 0:9] = a + d + c
quad[:, 9:15] = [0, 0, 1, 1, 1, 0]
quad[:, 15:24] = [0, 1, 0] * 3
with capi.Context(0) as c2:
    for m in scene.meshes[:2]:
        c2.upload_host_mesh(m)
    c2.upload_soup(quad, host_api.mesh_texture(scene.meshes[2]))
    out["ground_as_two_triangles"] = timeit(c2, scene.camera, scene.light, scene.transforms)
with capi.Context(0) as c3:
    # ground only (no drone, no treasure)
    c3.upload_host_mesh(scene.meshes[2])
    out["ground_alone"] = timeit(c3, scene.camera, scene.light, lambda f: scene.transforms(f)[2:])
    c3.clear_scene()
    c3.upload_soup(quad, host_api.mesh_texture(scene.meshes[2]))
    out["two_triangles_alone"] = timeit(c3, scene.camera, scene.light, lambda f: scene.transforms(f)[2:])
for name, keep in (("without_drone", (1, 2)), ("without_treasure", (0, 2)), ("drone_alone", (0,)), ("treasure_alone", (1,))):
    with capi.Context(0) as c4:
        for k in keep:
            c4.upload_host_mesh(scene.meshes[k])
        out[name] = timeit(c4, scene.camera, scene.light, lambda f: [scene.transforms(f)[k] for k in keep])
print(json.dumps(out))
scene.close()
