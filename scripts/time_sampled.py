#!/usr/bin/env python3
"""Kernel time of the sampled 4K frame (3840x2160, 16 jittered rays per pixel, 2 lights; frames 0 and 45) for build-against-build
comparisons (AB_CMD=scripts/time_sampled.py scripts/ab_kernels.sh)."""
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
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from raytracer_amd import capi, demo, host_api as H
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    for opt in sys.argv[3:]:
        k, v = opt.split("=")
        ctx.set_option(k, int(v))
    L = H.lib()
    lights = []
    for d, c in (((1, 1, -1), (1.4, 1.4, 1.4)), ((-1, 2, 0.5), (0.3, 0.5, 0.9))):
        l = L.create_directional_light(H.Vec3(*d), H.Vec3(*c))
        lights.append(capi.Light(capi.Vec3(l.direction.x, l.direction.y, l.direction.z), capi.Vec3(l.color.x, l.color.y, l.color.z)))
    W, Hh = 3840, 2160
    rgb = torch.empty((Hh, W, 3), dtype=torch.uint8, device="cuda:0")
    out = {"tag": tag}
    sha = hashlib.sha256()
    for f in (0, 45):
        ms = []
        for _ in range(passes + 1):
            ctx.render_sampled(scene.camera, lights, scene.transforms(f), W, Hh, 16, 0, rgb)
            ctx.sync()
            ms.append(ctx.last_kernel_ms())
        out[f"frame{f}_ms"] = round(float(np.median(ms[1:])), 4)
        sha.update(rgb.cpu().numpy().tobytes())
    out["sha"] = sha.hexdigest()[:16]
    print(json.dumps(out), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
