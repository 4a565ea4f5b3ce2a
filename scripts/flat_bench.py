#!/usr/bin/env python3
"""The stress scene as ONE mesh: side x side copies of the treasure baked into a single triangle soup
(about ten million distinct triangles), its BVH built on the GPU (rt_bvh_build_device), uploaded and
rendered with the plain frame loop next to the ground and the drone.  Unlike the instanced form of the
scene this working set (nodes + triangles + attributes, ~1.3 GB) does not fit any cache: the traversal
reads HBM.  One JSON line.  (tests/test_device_build.py compares such a baked scene with the CPU oracle.)

    python scripts/flat_bench.py --side 56 --frames 0,45
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def baked_grid(soup, side, spacing, seed):
    rng = np.random.default_rng(seed)
    n = soup.shape[0]
    out = np.empty((side * side * n, 24), np.float32)
    half = (side - 1) / 2.0
    k = 0
    for gx in range(side):
        for gz in range(side):
            ang = np.float32(rng.uniform(0, 6.28))
            c, s = np.float32(np.cos(ang)), np.float32(np.sin(ang))
            rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
            off = np.array([(gx - half) * spacing, 0.45 + 0.1 * rng.uniform(), (gz - half) * spacing], np.float32)
            blk = out[k * n:(k + 1) * n]
            blk[:] = soup
            for v in range(3):
                blk[:, 3 * v:3 * v + 3] = soup[:, 3 * v:3 * v + 3] @ rot.T + off
                blk[:, 15 + 3 * v:18 + 3 * v] = soup[:, 15 + 3 * v:18 + 3 * v] @ rot.T
            k += 1
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--side", type=int, default=56)
    p.add_argument("--spacing", type=float, default=1.05)
    p.add_argument("--frames", default="0,45")
    p.add_argument("--reps", type=int, default=5)
    p.add_argument("--host-build", action="store_true")
    p.add_argument("--no-regroup", action="store_true", help="host build: keep only the reference's tree (option regroup = 0)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--profile", action="store_true", help="no work-counting launches: the program a profiler is pointed at")
    a = p.parse_args()
    from raytracer_amd import capi, demo, host_api
    import side_roofline
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = demo.DemoScene(0)                                    # for the camera, the light, the animation and the assets
    treasure = host_api.mesh_triangles(scene.meshes[1])
    t0 = time.perf_counter()
    soup = baked_grid(treasure, a.side, a.spacing, 3)
    t_bake = time.perf_counter() - t0
    ctx = capi.Context(0)
    rgba = [host_api.mesh_texture(m) for m in scene.meshes]
    ctx.upload_host_mesh(scene.meshes[0])
    if a.host_build:
        ctx.set_option("device_build_min", 1 << 30)
    if a.no_regroup:
        ctx.set_option("regroup", 0)
    t0 = time.perf_counter()
    ctx.upload_soup(soup, rgba[1])                              # BVH build + packing + upload in one call
    t_build = time.perf_counter() - t0
    stats = {} if a.host_build else ctx.build_stats()
    n_tris = int(soup.shape[0])
    ctx.upload_host_mesh(scene.meshes[2])
    W, H = a.width, a.height
    rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    rows = []
    for f in [int(x) for x in a.frames.split(",")]:
        tfs = [scene.transforms(f)[0], capi.Transform(), capi.Transform()]
        c = {"primary_rays": W * H, "shadow_rays": 0, "slab_tests": 0, "tri_tests": 0}
        if not a.profile:
            ctx.render(scene.camera, scene.light, tfs, W, H, rgb, flags=capi.RT_FLAG_COUNT_WORK)
            ctx.sync()
            c = ctx.last_counters()
        ms = []
        for _ in range(a.reps):
            ctx.render(scene.camera, scene.light, tfs, W, H, rgb)
            ctx.sync()
            ms.append(ctx.last_kernel_ms())
        t = float(np.median(ms))
        # the timed call stores the colour only and sends no shadow ray that cannot change it (max(0.2, n.L) = 0.2 already);
        # the counting call above sends them and says how many: they are left out of the rate
        rays = c["primary_rays"] + c["shadow_rays"] - c.get("idle_shadow_rays", 0)
        rows.append({"frame": f, "kernel_ms": round(t, 3), "rays": rays, "shadow_rays_not_sent": c.get("idle_shadow_rays", 0),
                     "Mrays_per_s_counting_the_reference_rays": round((rays + c.get("idle_shadow_rays", 0)) / (t * 1e-3) / 1e6, 1), "Mrays_per_s": round(rays / (t * 1e-3) / 1e6, 1),
                     "box_tests_per_ray": round(c["slab_tests"] / rays, 1), "triangle_tests_per_ray": round(c["tri_tests"] / rays, 1)})
        if not a.profile:
            rows[-1]["roofline"] = side_roofline.roofline("render_tiles (rt_render_frame's default kernel), ten-million-triangle mesh", c, rays * 3, t, "r3_pmc_flat.json")
    cpu = None
    if not a.profile and not a.no_cpu_baseline:
        # the reference's loop over a smaller scene of the same kind (13 x 13 copies, 533 k triangles): its own build of that soup
        from oracle import pyoracle as po
        po.lib()
        om = po.default_meshes()
        small = baked_grid(treasure, 13, a.spacing, 3)
        bt, bn = po.build_bvh(small)
        meshes3 = [om[0], po.HostMesh(bt, bn, om[1].rgba), om[2]]
        f0 = int(a.frames.split(",")[0])
        otf = [po.animation_transforms(f0)[0], po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0)), po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]

        def run():
            fr = po.render_frame(meshes3, otf, po.default_camera(), po.default_light(), 480, 270)
            return fr.counters["primary_rays"] + fr.counters["shadow_rays"], po.lib().orc_max_threads()
        cpu = side_roofline.cpu_baseline(run, "13 x 13 copies baked into one mesh of 533 195 triangles (the full scene's build alone is minutes of CPU), 480x270, "
                                              "the reference's un-culled walk (bvh.c:80-129), OpenMP dynamic rows")
    out = {"metric": "Mrays/s (primary+shadow), stress scene as one mesh", "unit": "Mrays/s",
           "value": round(float(np.mean([r["Mrays_per_s"] for r in rows])), 1),
           "config": {"workload": f"{W}x{H}, {a.side}x{a.side} treasures baked into one mesh of {n_tris} triangles + ground + drone"},
           "dtype": "f32", "bake_s": round(t_bake, 2), "build_pack_upload_s": round(t_build, 3), "bvh_build": "host" if a.host_build else "device",
           "build_stats": stats, "frames": rows, "roofline": rows[0].get("roofline"), "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)
    ctx.close()
    scene.close()


if __name__ == "__main__":
    main()
