#!/usr/bin/env python3
"""Kernel time of the instanced stress frame (scripts/instanced_bench.py's scene, frames 0 and 45, no per-frame rebuild) for
build-against-build comparisons (AB_CMD=scripts/time_instanced.py scripts/ab_kernels.sh)."""
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    from raytracer_amd import capi, demo
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    for opt in sys.argv[3:]:
        k, v = opt.split("=")
        ctx.set_option(k, int(v))
    rng = np.random.default_rng(3)
    side, spacing = 56, 1.05
    ids, tfs = [0, 2], [None, capi.Transform()]
    half = (side - 1) / 2.0
    for gx in range(side):
        for gz in range(side):
            ids.append(1)
            tfs.append(capi.Transform(capi.Vec3((gx - half) * spacing, 0.45 + 0.1 * rng.uniform(), (gz - half) * spacing),
                                      capi.Vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
    W, H = 1920, 1080
    rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    out = {"tag": tag}
    sha = hashlib.sha256()
    for f in (0, 45):
        tfs[0] = scene.transforms(f)[0]
        inst = capi.Context.make_instances(ids, tfs)
        ms = []
        for _ in range(passes + 2):
            ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb)
            ctx.sync()
            ms.append(ctx.last_kernel_ms())
        out[f"frame{f}_ms"] = round(float(np.median(ms[2:])), 4)
        sha.update(rgb.cpu().numpy().tobytes())
    out["sha"] = sha.hexdigest()[:16]
    print(json.dumps(out), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
