#!/usr/bin/env python3
"""The instanced stress frame (56 x 56 treasures + ground + drone) as it is and with every treasure replaced by an octahedron
in its box: what is left then is the hierarchy over the instances and the per-pixel fixed part.  Prints kernel ms and box /
triangle tests per ray.  Run on a GPU: python scripts/cost_split_instanced.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import capi, demo, host_api
scene = demo.DemoScene(0)
W, H = 1920, 1080
rng = np.random.default_rng(3)
ids, tfs = [0, 2], [scene.transforms(0)[0], capi.Transform()]
for gx in range(56):
    for gz in range(56):
        ids.append(1)
        tfs.append(capi.Transform(capi.Vec3((gx - 27.5) * 1.05, 0.45 + 0.1 * rng.uniform(), (gz - 27.5) * 1.05), capi.Vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
def timeit(ctx):
    inst = capi.Context.make_instances(ids, tfs)
    ms = []
    for _ in range(6):
        ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb)
        ms.append(ctx.last_kernel_ms())
    ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb, flags=capi.RT_FLAG_COUNT_WORK); ctx.sync()
    c = ctx.last_counters()
    rays = c["primary_rays"] + c["shadow_rays"]
    return {"ms": round(float(np.median(ms)), 3), "slab_per_ray": round(c["slab_tests"] / rays, 1), "tri_per_ray": round(c["tri_tests"] / rays, 1)}
out = {"normal": timeit(scene.ctx)}
# treasure replaced by a small octahedron inside its box (walk inside an instance costs next to nothing)
t = host_api.mesh_triangles(scene.meshes[1])
p = t[:, 0:9].reshape(-1, 3); lo, hi = p.min(axis=0), p.max(axis=0); c = (lo + hi) / 2; r = (hi - lo) / 2
V = [c + [r[0], 0, 0], c - [r[0], 0, 0], c + [0, r[1], 0], c - [0, r[1], 0], c + [0, 0, r[2]], c - [0, 0, r[2]]]
F = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
oct_ = np.zeros((8, 24), np.float32)
for k, (a, b, d) in enumerate(F):
    oct_[k, 0:9] = np.concatenate([V[a], V[b], V[d]]); oct_[k, 9:15] = [0, 0, 1, 0, 0, 1]; oct_[k, 15:24] = [0, 1, 0] * 3
with capi.Context(0) as c2:
    c2.upload_host_mesh(scene.meshes[0]); c2.upload_soup(oct_, host_api.mesh_texture(scene.meshes[1])); c2.upload_host_mesh(scene.meshes[2])
    out["treasures_as_octahedra"] = timeit(c2)
print(json.dumps(out))
scene.close()
