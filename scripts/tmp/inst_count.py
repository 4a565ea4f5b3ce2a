import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracer_amd import capi, demo
scene = demo.DemoScene(0); ctx = scene.ctx
rng = np.random.default_rng(3)
ids, tfs = [0, 2], [scene.transforms(0)[0], capi.Transform()]
for gx in range(56):
    for gz in range(56):
        ids.append(1)
        tfs.append(capi.Transform(capi.Vec3((gx - 27.5) * 1.05, 0.45 + 0.1 * rng.uniform(), (gz - 27.5) * 1.05), capi.Vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
rgb = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
inst = capi.Context.make_instances(ids, tfs)
ctx.render_instanced(scene.camera, scene.light, inst, 1920, 1080, rgb, flags=capi.RT_FLAG_COUNT_WORK)
ctx.sync()
print(json.dumps(ctx.last_counters()))
scene.close()
