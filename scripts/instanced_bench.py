#!/usr/bin/env python3
"""Times the instanced stress configuration: the treasure mesh placed on a seeded grid (about ten
million placed triangles), the ground and the moving drone, the instance hierarchy rebuilt for every
frame (rt_render_frame_instanced).  Prints one JSON line.

    python scripts/instanced_bench.py --side 56 --frames 0,45
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


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--side", type=int, default=56, help="treasures per side of the grid")
    p.add_argument("--spacing", type=float, default=1.05)
    p.add_argument("--frames", default="0,45")
    p.add_argument("--reps", type=int, default=5)
    p.add_argument("--wave", type=int, default=-1, help="option instanced_wave: tiles per wave of the per-wavefront scheduler, 0 = nested loops per lane")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-rebuild", action="store_true", help="move the drone by its transform only (no per-frame BVH rebuild)")
    a = p.parse_args()
    from raytracer_amd import capi, demo
    import side_roofline
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    if a.wave >= 0:
        ctx.set_option("instanced_wave", a.wave)
    rng = np.random.default_rng(3)
    ids, tfs = [0, 2], [None, capi.Transform()]
    half = (a.side - 1) / 2.0
    for gx in range(a.side):
        for gz in range(a.side):
            ids.append(1)
            tfs.append(capi.Transform(capi.Vec3((gx - half) * a.spacing, 0.45 + 0.1 * rng.uniform(), (gz - half) * a.spacing),
                                      capi.Vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
    tri_counts = [int(m.triangle_count) for m in scene.meshes]
    placed = sum(tri_counts[i] for i in ids)
    W, H = a.width, a.height
    rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    rows = []
    drone_soup = None
    if not a.no_rebuild:
        from raytracer_amd import host_api
        drone_soup = host_api.mesh_triangles(scene.meshes[0])        # [n, 24] float32, as create_mesh left them

    def place_drone(f):
        """Either the drone's transform, or its vertices rewritten on the host, its BVH rebuilt and swapped in."""
        if drone_soup is None:
            tfs[0] = scene.transforms(f)[0]
            return
        t = scene.transforms(f)[0]
        ang = np.float32(t.rotation.y)
        c, s = np.float32(np.cos(ang)), np.float32(np.sin(ang))
        rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
        off = np.array([t.position.x, t.position.y, t.position.z], np.float32)
        baked = drone_soup.copy()
        for k in range(3):
            baked[:, 3 * k:3 * k + 3] = drone_soup[:, 3 * k:3 * k + 3] @ rot.T + off
            baked[:, 15 + 3 * k:18 + 3 * k] = drone_soup[:, 15 + 3 * k:18 + 3 * k] @ rot.T
        tris, nodes = capi.build_bvh(baked)
        ctx.replace_mesh(0, tris, nodes)
        tfs[0] = capi.Transform()

    for f in [int(x) for x in a.frames.split(",")]:
        place_drone(f)                                               # the drone moves, everything else stays
        inst = capi.Context.make_instances(ids, tfs)
        ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb, flags=capi.RT_FLAG_COUNT_WORK)
        ctx.sync()
        c = ctx.last_counters()
        ms, wall = [], []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            place_drone(f)
            ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb)
            ctx.sync()
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(ctx.last_kernel_ms())
        t = float(np.median(ms))
        # the timed call stores the colour only and sends no shadow ray that cannot change it (max(0.2, n.L) = 0.2 already);
        # the counting call above sends them and says how many: they are left out of the rate
        rays = c["primary_rays"] + c["shadow_rays"] - c.get("idle_shadow_rays", 0)
        # the same frame with the fast build of the kernels (option "fast": fused arithmetic, the tree over the triangles
        # themselves): not the default, not in `value`; how far its image is from the exact one is stated beside its time
        hid = torch.empty((H, W), dtype=torch.int32, device=dev)
        htri = torch.empty((H, W), dtype=torch.int32, device=dev)
        ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb, hit_instance=hid, hit_tri=htri)
        ctx.sync()
        exact = (rgb.cpu().numpy().astype(np.int16), hid.cpu().numpy(), htri.cpu().numpy())
        ctx.set_option("fast", 1)
        ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb, hit_instance=hid, hit_tri=htri, flags=capi.RT_FLAG_COUNT_WORK)
        ctx.sync()
        cf = ctx.last_counters()
        fms = []
        for _ in range(a.reps):
            ctx.render_instanced(scene.camera, scene.light, inst, W, H, rgb, hit_instance=hid, hit_tri=htri)
            ctx.sync()
            fms.append(ctx.last_kernel_ms())
        ctx.set_option("fast", 0)
        fast_rgb, fast_hid, fast_tri = rgb.cpu().numpy().astype(np.int16), hid.cpu().numpy(), htri.cpu().numpy()
        fast = {"kernel_ms": round(float(np.median(fms)), 3), "speedup_over_exact_kernel": round(t / float(np.median(fms)), 3),
                "same_hit_share": round(float(((fast_hid == exact[1]) & (fast_tri == exact[2])).mean()), 7),
                "rgb_within_1_share": round(float((np.abs(fast_rgb - exact[0]).max(axis=2) <= 1).mean()), 7),
                "box_tests": cf["slab_tests"], "triangle_tests": cf["tri_tests"], "box_tests_exact": c["slab_tests"], "triangle_tests_exact": c["tri_tests"]}
        rows.append({"frame": f, "kernel_ms": round(t, 3), "call_ms_with_host_rebuild": round(float(np.median(wall)), 3),
                     "rays": rays, "shadow_rays_not_sent": c.get("idle_shadow_rays", 0),
                     "Mrays_per_s_counting_the_reference_rays": round((rays + c.get("idle_shadow_rays", 0)) / (t * 1e-3) / 1e6, 1), "Mrays_per_s": round(rays / (t * 1e-3) / 1e6, 1),
                     # ray transforms are not counted by the kernel: two proposed instances per ray is what the hierarchy gives on this scene
                     "roofline": side_roofline.roofline("render_instanced (block 64)", c, rays * 2, t, "r3_pmc_instanced.json"),
                     "fast_build": fast})
    cpu = None
    if not a.no_cpu_baseline:
        from oracle import pyoracle as po
        po.lib()
        om = po.default_meshes()
        otf = [po.Transform(po.vec3(t.position.x, t.position.y, t.position.z), po.vec3(t.rotation.x, t.rotation.y, t.rotation.z)) for t in tfs[1:]]
        otf.insert(0, po.animation_transforms(int(a.frames.split(",")[0]))[0])           # the drone placed by its transform

        def run():
            fr, _, _ = po.render_frame_ids([om[i] for i in ids], otf, po.default_camera(), po.default_light(), 48, 27)
            return fr.counters["primary_rays"] + fr.counters["shadow_rays"], po.lib().orc_max_threads()
        cpu = side_roofline.cpu_baseline(run, f"the same {len(ids)}-entry list at 48x27 through the reference's loop over the whole list (scene.c:72-108), OpenMP dynamic rows")
    out = {"metric": "Mrays/s (primary+shadow), instanced scene", "unit": "Mrays/s",
           "value": round(float(np.mean([r["Mrays_per_s"] for r in rows])), 1),
           "config": {"workload": f"{W}x{H}, {len(ids)} instances ({placed} placed triangles): {a.side}x{a.side} treasures + ground + drone, "
                                  "hierarchy over the instances rebuilt on the host per frame"
                                  + ("" if a.no_rebuild else "; the drone's vertices rewritten, its BVH rebuilt and re-uploaded per frame")},
           "dtype": "f32", "instanced_wave": ctx.get_option("instanced_wave"), "frames": rows, "roofline": rows[0]["roofline"], "cpu_baseline": cpu,
           "fast_build": {**rows[0]["fast_build"], "note": "option \"fast\" (not the default, not in `value`): frame " + str(rows[0]["frame"])}}
    print(json.dumps(out), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
