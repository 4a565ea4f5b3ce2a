#!/usr/bin/env python3
"""A long random run of the parity check the GPU tests make on six scenes: the reference driver's three meshes under random
placements (anywhere within a few units, any rotation, the ground tilted), cameras anywhere around and inside them (looking at a
random point, field of view 15..120 degrees), random light directions, frame sizes that are not multiples of the tile; every
plane of the HIP path (rgb, hit id, bits of the closest t, shadow flag) against the CPU oracle, which is test infrastructure.
Prints one JSON line; exits non-zero on the first scene that differs and says which.

    python scripts/soak_parity.py --scenes 1000 --seed 1
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402


def instanced(a):
    import torch
    from oracle import pyoracle as po
    from raytracer_amd import capi
    po.lib()
    meshes = po.default_meshes()
    ctx = capi.Context(0)
    for m in meshes:
        ctx.upload_mesh(m.tris, m.nodes, m.rgba)
    rng = np.random.default_rng(a.seed)
    sizes = [(160, 90), (97, 61), (200, 40)]
    pixels = hits = 0
    dev = "cuda:0"
    special = [0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e30, -1e30, 1e-30, 1e-42, 1e-7, -1e-7, 1.0, 90.0, 180.0, 360.0]

    def wild(values):
        return tuple(float(rng.choice(special)) if rng.uniform() < a.wild else float(v) for v in values)

    prev_view = None
    for k in range(a.scenes):
        if a.mutate and k % 40 != 0:
            # the previous list with a few entries moved, swapped for another mesh, or the list one entry longer or shorter:
            # the kept hierarchy is patched, placement constants of the others are reused
            ids, tfs = list(ids), list(tfs)
            for _ in range(int(rng.integers(1, 5))):
                j = int(rng.integers(len(ids)))
                what = rng.integers(4)
                if what == 0:
                    ids[j] = int(rng.choice([0, 1, 2]))
                elif what == 1 and len(ids) > 2:
                    del ids[j]; del tfs[j]
                elif what == 2 and len(ids) < 60:
                    ids.append(int(rng.choice([0, 1, 2])))
                    tfs.append(po.Transform(po.vec3(*wild(rng.uniform(-spread, spread, 3))), po.vec3(*wild(rng.uniform(-3.2, 3.2, 3)))))
                else:
                    tfs[j] = po.Transform(po.vec3(*wild(rng.uniform(-spread, spread, 3))), po.vec3(*wild(rng.uniform(-3.2, 3.2, 3))))
            n = len(ids)
        else:
            n = int(rng.integers(1, 41))
            spread = float(rng.choice([1.0, 3.0, 8.0]))
            ids = [int(rng.choice([0, 1, 1, 1, 2])) for _ in range(n)]
            tfs = [po.Transform(po.vec3(*wild(rng.uniform(-spread, spread, 3))), po.vec3(*wild(rng.uniform(-3.2, 3.2, 3)))) for _ in range(n)]
        if n > 3 and rng.uniform() < 0.3:                       # coincident entries: the earlier one wins
            tfs[n - 1] = tfs[0]; ids[n - 1] = ids[0]
        eye = rng.uniform(-1.2 * spread - 2, 1.2 * spread + 2, 3)
        at = rng.uniform(-spread, spread, 3)
        if np.linalg.norm(eye - at) < 1e-3:
            at = eye + np.array([0.3, -0.2, 0.9])
        fov = wild([rng.uniform(20, 110)])[0]
        ocam = po.Camera(po.vec3(*wild(eye)), po.vec3(*wild(at)), po.vec3(*wild((0, 1, 0))), fov)
        if a.wild > 0 and rng.uniform() < 0.5:
            olight = po.Light(po.vec3(*wild(rng.normal(size=3))), po.vec3(*wild(rng.uniform(-0.5, 2.0, 3))))
        else:
            olight = po.lib().orc_make_light(po.vec3(*rng.normal(size=3)), po.vec3(*rng.uniform(0.3, 1.6, 3)))
        cam = capi.Camera(capi.Vec3(*ocam.position.tup()), capi.Vec3(*ocam.look_at.tup()), capi.Vec3(*ocam.up.tup()), fov)
        light = capi.Light(capi.Vec3(*olight.direction.tup()), capi.Vec3(*olight.color.tup()))
        if a.mutate and k % 40 != 0 and prev_view is not None and rng.uniform() < 0.7:
            ocam, olight, cam, light = prev_view                # same camera and light as the step before: constants are reused
        prev_view = (ocam, olight, cam, light)
        w, h = sizes[k % len(sizes)]
        fr, inst, tri = po.render_frame_ids([meshes[i] for i in ids], tfs, ocam, olight, w, h)
        lst = capi.Context.make_instances(ids, [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs])
        rgb = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
        hi = torch.empty((h, w), dtype=torch.int32, device=dev)
        htri = torch.empty((h, w), dtype=torch.int32, device=dev)
        t = torch.empty((h, w), dtype=torch.float32, device=dev)
        sh = torch.empty((h, w), dtype=torch.uint8, device=dev)
        ctx.render_instanced(cam, light, lst, w, h, rgb, hit_instance=hi, hit_tri=htri, hit_t=t, shadow=sh)
        ctx.sync()
        same = (np.array_equal(hi.cpu().numpy(), inst) and np.array_equal(htri.cpu().numpy(), tri)
                and np.array_equal(t.cpu().numpy().view(np.uint32), fr.hit_t.view(np.uint32))
                and np.array_equal(sh.cpu().numpy(), fr.shadow) and np.array_equal(rgb.cpu().numpy(), fr.rgb))
        if not same:
            print(json.dumps({"scene": k, "seed": a.seed, "instances": n, "size": [w, h], "differs": True, "ids": ids,
                              "placements": [[repr(v) for v in t.position.tup() + t.rotation.tup()] for t in tfs],
                              "camera": [repr(v) for v in ocam.position.tup() + ocam.look_at.tup() + ocam.up.tup()] + [repr(fov)],
                              "light": [repr(v) for v in olight.direction.tup() + olight.color.tup()]}))
            sys.exit(1)
        pixels += w * h
        hits += int((inst >= 0).sum())
        if (k + 1) % 100 == 0:
            print(f"[soak] {k + 1} instanced scenes identical", file=sys.stderr, flush=True)
    print(json.dumps({"scenes": a.scenes, "seed": a.seed, "path": "rt_render_frame_instanced, lists of 1..40 placed meshes", "pixels": pixels,
                      "pixels_with_a_hit": hits, "planes": "rgb, instance, triangle, closest t (bits), shadow", "identical": True, "wild": a.wild}))
    ctx.close()


def soups(a):
    """Random triangle soups (1..400 triangles, sometimes flat, sometimes many identical) whose vertex, texture and normal values are
    replaced with probability --wild by special values: built on the host (against the oracle's build, bit for bit) and on the
    device, packed on the host and on the device, rendered in both traversal modes against the oracle's reference loop."""
    import torch  # noqa: F401
    from oracle import pyoracle as po
    from raytracer_amd import capi
    po.lib()
    rng = np.random.default_rng(a.seed)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 1e-30, 1e-42, 1e-7, -1e-7, 1.0, 3e38, -3e38], np.float32)
    tex = rng.integers(0, 256, size=(8, 8, 4), dtype=np.uint8)
    l = po.default_light()
    light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    pixels = hits = 0
    w, h = 96, 54
    for k in range(a.scenes):
        n = int(rng.choice([1, 2, 5, 17, 64, 150, 400]))
        soup = (rng.normal(size=(n, 24)) * rng.choice([0.3, 1.0, 3.0])).astype(np.float32)
        kind = rng.integers(4)
        if kind == 1:
            soup[:, [1, 4, 7]] = np.float32(0.25)                       # flat in y
        elif kind == 2:
            soup[n // 3:, :9] = soup[0, :9]                               # many identical triangles: partitions that fail
        mask = rng.uniform(size=soup.shape) < a.wild
        soup[mask] = rng.choice(special, size=int(mask.sum()))
        # vertices get no finite giants: with coordinates of 1e30 .. 3e38 the products of ray.c:46-62 overflow, u comes out as
        # inf - inf = NaN, passes every comparison (they are all false), and the reference accepts the hit with a finite t --
        # and then indexes its texture with (int)NaN.  NaN and infinite coordinates never get that far (t is NaN or 0).
        pos = soup[:, :9]
        giant = np.isfinite(pos) & (np.abs(pos) > 1e20)
        pos[giant] = np.float32(np.inf)
        # texture coordinates stay finite: with a NaN or infinite one the reference computes (int)(NaN * (size - 1)) and reads
        # the texture far out of bounds (mesh.c:143-147) -- there is no result to compare with
        uv = soup[:, 9:15]
        uv[~np.isfinite(uv)] = np.float32(1e30)
        t, nodes = capi.build_bvh(soup)
        wt, wn = po.build_bvh(soup)
        same = np.array_equal(t.view(np.uint32), wt.view(np.uint32)) and nodes.tobytes() == wn.tobytes()
        tf = po.Transform(po.vec3(*rng.uniform(-0.5, 0.5, 3)), po.vec3(*rng.uniform(-3.2, 3.2, 3)))
        ctf = [capi.Transform(capi.Vec3(*tf.position.tup()), capi.Vec3(*tf.rotation.tup()))]
        what = "host build"
        if same:
            want = po.render_frame([po.HostMesh(wt, wn, tex)], [tf], po.default_camera(), l, w, h)
            for route in ("host", "device"):
                with capi.Context(0) as c:
                    if route == "host":
                        c.upload_mesh(t, nodes, tex)
                    else:
                        c.set_option("device_build_min", 1)
                        _, post = c.upload_soup(soup, tex, want_triangles=True)
                        same = same and np.array_equal(post.view(np.uint32), wt.view(np.uint32))
                    for flags in (0, capi.RT_FLAG_REFERENCE_ORDER):
                        rgb, hid, ht, sh = c.render_host(cam, light, ctf, w, h, flags=flags)
                        same = (same and np.array_equal(hid, want.hit_id) and np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32))
                                and np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb))
                if not same:
                    what = f"{route} route"
                    break
        if not same:
            np.save(os.path.join(REPO, "gpurun_out", f"soup_fail_{a.seed}_{k}.npy"), soup)
            print(json.dumps({"scene": k, "seed": a.seed, "triangles": n, "differs": what,
                              "placement": [repr(v) for v in tf.position.tup() + tf.rotation.tup()]}))
            sys.exit(1)
        pixels += w * h
        hits += int((want.hit_id >= 0).sum())
        if (k + 1) % 100 == 0:
            print(f"[soak] {k + 1} soups identical", file=sys.stderr, flush=True)
    print(json.dumps({"scenes": a.scenes, "seed": a.seed, "path": "random soups: rt_bvh_build, rt_mesh_upload / rt_mesh_upload_soup (device build and "
                      "packing), rt_render_frame in both traversal modes", "pixels": pixels, "pixels_with_a_hit": hits, "identical": True, "wild": a.wild}))


def multi(a):
    """Scenes of one to six random soups (as in --soups) under random placements, cameras and lights: between-mesh ties, screen and
    shadow rectangles of boxes with infinite or NaN bounds, scene bounds that are infinite.  rt_render_frame in both traversal
    modes, rt_render_frames, and the same list through rt_render_frame_instanced, against the oracle's reference loop."""
    import torch
    from oracle import pyoracle as po
    from raytracer_amd import capi
    po.lib()
    rng = np.random.default_rng(a.seed)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 1e-30, 1e-42, 1e-7, -1e-7, 1.0, 3e38, -3e38], np.float32)
    tex = rng.integers(0, 256, size=(8, 8, 4), dtype=np.uint8)
    pixels = hits = 0
    w, h = 96, 54
    for k in range(a.scenes):
        m = int(rng.integers(1, 7))
        built, tfs = [], []
        for _ in range(m):
            n = int(rng.choice([1, 2, 5, 17, 64, 150]))
            soup = (rng.normal(size=(n, 24)) * rng.choice([0.3, 1.0, 3.0])).astype(np.float32)
            kind = rng.integers(4)
            if kind == 1:
                soup[:, [1, 4, 7]] = np.float32(0.25)
            elif kind == 2 and rng.uniform() < 0.5:
                soup[:, :9] = built[-1][0][: n, :9] if built and built[-1][0].shape[0] >= n else soup[:, :9]      # coincident with the previous mesh
            mask = rng.uniform(size=soup.shape) < a.wild
            soup[mask] = rng.choice(special, size=int(mask.sum()))
            uv = soup[:, 9:15]
            uv[~np.isfinite(uv)] = np.float32(1e30)
            pos = soup[:, :9]
            pos[np.isfinite(pos) & (np.abs(pos) > 1e20)] = np.float32(np.inf)
            built.append(po.build_bvh(soup))
            same_place = tfs and rng.uniform() < 0.2
            tfs.append(tfs[-1] if same_place else po.Transform(po.vec3(*rng.uniform(-1.5, 1.5, 3)), po.vec3(*rng.uniform(-3.2, 3.2, 3))))
        eye = rng.uniform(-4, 4, 3)
        ocam = po.Camera(po.vec3(*eye), po.vec3(*rng.uniform(-1, 1, 3)), po.vec3(0, 1, 0), float(rng.uniform(20, 110)))
        olight = po.lib().orc_make_light(po.vec3(*rng.normal(size=3)), po.vec3(*rng.uniform(0.3, 1.6, 3)))
        cam = capi.Camera(capi.Vec3(*ocam.position.tup()), capi.Vec3(*ocam.look_at.tup()), capi.Vec3(0, 1, 0), ocam.fov)
        light = capi.Light(capi.Vec3(*olight.direction.tup()), capi.Vec3(*olight.color.tup()))
        meshes = [po.HostMesh(t, nd, tex) for t, nd in built]
        _, inst, tri = po.render_frame_ids(meshes, tfs, ocam, olight, w, h)
        fr = po.render_frame(meshes, tfs, ocam, olight, w, h)
        ctf = [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs]
        what = None
        with capi.Context(0) as c:
            for t, nd in built:
                c.upload_mesh(t, nd, tex)
            for flags in (0, capi.RT_FLAG_REFERENCE_ORDER):
                rgb, hid, ht, sh = c.render_host(cam, light, ctf, w, h, flags=flags)
                if not (np.array_equal(hid, fr.hit_id) and np.array_equal(ht.view(np.uint32), fr.hit_t.view(np.uint32))
                        and np.array_equal(sh, fr.shadow) and np.array_equal(rgb, fr.rgb)):
                    what = what or f"rt_render_frame flags {flags}"
            two = torch.empty((2, h, w, 3), dtype=torch.uint8, device="cuda:0")
            ins = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
            hi = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
            htri = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
            c.render_frames(cam, light, [ctf, ctf], w, h, two)
            c.render_instanced(cam, light, capi.Context.make_instances(list(range(m)), ctf), w, h, ins, hit_instance=hi, hit_tri=htri)
            c.sync()
            if what is None and not (np.array_equal(two[0].cpu().numpy(), fr.rgb) and np.array_equal(two[1].cpu().numpy(), fr.rgb)):
                what = "rt_render_frames"
            if what is None and not (np.array_equal(ins.cpu().numpy(), fr.rgb) and np.array_equal(hi.cpu().numpy(), inst) and np.array_equal(htri.cpu().numpy(), tri)):
                what = "rt_render_frame_instanced"
        if what:
            np.savez(os.path.join(REPO, "gpurun_out", f"multi_fail_{a.seed}_{k}.npz"), **{f"tris{i}": b[0] for i, b in enumerate(built)},
                     **{f"nodes{i}": b[1] for i, b in enumerate(built)})
            print(json.dumps({"scene": k, "seed": a.seed, "meshes": m, "differs": what,
                              "placements": [[repr(v) for v in t.position.tup() + t.rotation.tup()] for t in tfs],
                              "camera": [repr(v) for v in ocam.position.tup() + ocam.look_at.tup()] + [repr(ocam.fov)],
                              "light": [repr(v) for v in olight.direction.tup() + olight.color.tup()]}))
            sys.exit(1)
        pixels += w * h
        hits += int((fr.hit_id >= 0).sum())
        if (k + 1) % 100 == 0:
            print(f"[soak] {k + 1} scenes of several soups identical", file=sys.stderr, flush=True)
    print(json.dumps({"scenes": a.scenes, "seed": a.seed, "path": "one to six random soups per scene: rt_render_frame in both traversal modes, rt_render_frames, "
                      "rt_render_frame_instanced", "pixels": pixels, "pixels_with_a_hit": hits, "identical": True, "wild": a.wild}))


def rebuilds(a):
    """One context, two meshes (the reference's ground and one that keeps changing): every step gives the second mesh a new random
    soup of 0 .. 30 000 triangles through rt_mesh_rebuild (build and packing on the device or on the host, by the size threshold of
    the step) or through rt_bvh_build + rt_mesh_replace, then renders (plain frame, batched frames, instanced list) against the
    oracle's reference loop over the oracle's own build of the same soup.  Device arrays are reused or replaced as sizes come and go."""
    import torch
    from oracle import pyoracle as po
    from raytracer_amd import capi
    po.lib()
    rng = np.random.default_rng(a.seed)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, 1e-42, 1e-7, -1e-7, 1.0], np.float32)
    ground = po.default_meshes()[2]
    tex = rng.integers(0, 256, size=(8, 8, 4), dtype=np.uint8)
    w, h = 96, 54
    l = po.default_light()
    light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    pixels = hits = 0
    with capi.Context(0) as c:
        c.upload_mesh(ground.tris, ground.nodes, ground.rgba)
        c.upload_soup((rng.normal(size=(5, 24))).astype(np.float32), tex)
        for k in range(a.scenes):
            n = int(rng.choice([0, 1, 3, 4, 5, 50, 300, 300, 2000, 30000], p=[.05, .1, .1, .1, .1, .15, .15, .1, .1, .05]))
            soup = (rng.normal(size=(n, 24)) * rng.choice([0.3, 1.0])).astype(np.float32)
            soup[:, :9] = soup[:, :9] * np.float32(0.2) + np.repeat(rng.normal(size=(n, 3)).astype(np.float32) * np.float32(0.8), 3, axis=1).reshape(n, 9)[:, [0, 3, 6, 1, 4, 7, 2, 5, 8]]
            mask = rng.uniform(size=soup.shape) < a.wild
            soup[mask] = rng.choice(special, size=int(mask.sum()))
            uv = soup[:, 9:15]
            uv[~np.isfinite(uv)] = np.float32(1e30)
            wt, wn = po.build_bvh(soup)
            route = int(rng.integers(3))
            if route == 0:
                c.set_option("device_build_min", 1)
                post = c.rebuild_mesh(1, soup, want_triangles=True)
            elif route == 1:
                c.set_option("device_build_min", 1 << 30)
                post = c.rebuild_mesh(1, soup, want_triangles=True)
            else:
                post, nodes = capi.build_bvh(soup)
                c.replace_mesh(1, post, nodes)
            same = np.array_equal(np.asarray(post).view(np.uint32), wt.view(np.uint32))
            tfs = [po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0)), po.Transform(po.vec3(*rng.uniform(-0.5, 0.5, 3) + [0, 0.8, 0]), po.vec3(*rng.uniform(-3.2, 3.2, 3)))]
            ctf = [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs]
            want = po.render_frame([ground, po.HostMesh(wt, wn, tex)], tfs, po.default_camera(), l, w, h)
            rgb, hid, ht, sh = c.render_host(cam, light, ctf, w, h)
            two = torch.empty((2, h, w, 3), dtype=torch.uint8, device="cuda:0")
            ins = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
            c.render_frames(cam, light, [ctf, ctf], w, h, two)
            c.render_instanced(cam, light, capi.Context.make_instances([0, 1, 1], ctf + [ctf[1]]), w, h, ins)
            c.sync()
            same = (same and np.array_equal(hid, want.hit_id) and np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32))
                    and np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb) and np.array_equal(two[1].cpu().numpy(), want.rgb)
                    and np.array_equal(ins.cpu().numpy(), want.rgb))
            if not same:
                np.save(os.path.join(REPO, "gpurun_out", f"rebuild_fail_{a.seed}_{k}.npy"), soup)
                print(json.dumps({"step": k, "seed": a.seed, "triangles": n, "route": ["rt_mesh_rebuild on the device", "rt_mesh_rebuild on the host", "rt_mesh_replace"][route],
                                  "differs": True, "placement": [repr(v) for v in tfs[1].position.tup() + tfs[1].rotation.tup()]}))
                sys.exit(1)
            pixels += w * h
            hits += int((want.hit_id >= 0).sum())
            if (k + 1) % 100 == 0:
                print(f"[soak] {k + 1} rebuilds identical", file=sys.stderr, flush=True)
    print(json.dumps({"scenes": a.scenes, "seed": a.seed, "path": "one mesh rebuilt every step (rt_mesh_rebuild on device / host, rt_mesh_replace; 0..30 000 triangles), "
                      "rt_render_frame, rt_render_frames, rt_render_frame_instanced", "pixels": pixels, "pixels_with_a_hit": hits, "identical": True, "wild": a.wild}))


def verify_skips(a):
    """No oracle in the loop: the kernels check their own shortcuts (RT_FLAG_VERIFY_SKIPS).  Random scenes of three kinds -- the
    reference's meshes under random placements, cameras and lights (frame sizes up to 1280x720); one to four random soups, flat
    sheets among them (grid entries); lists of 1..40 placed meshes through rt_render_frame_instanced -- are rendered with the
    flag; every disagreement counter has to stay zero and the planes have to equal the same frame rendered without the flag."""
    import torch
    from oracle import pyoracle as po                # only for the meshes and for building BVHs of random soups
    from raytracer_amd import capi
    po.lib()
    meshes = po.default_meshes()
    ctx = capi.Context(0)
    for m in meshes:
        ctx.upload_mesh(m.tris, m.nodes, m.rgba)
    rng = np.random.default_rng(a.seed)
    special = [0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e30, -1e30, 1e-30, 1e-42, 1e-7, -1e-7, 1.0, 90.0, 180.0, 360.0]
    kinds = ("vis_disagreements", "shadow_vis_disagreements", "entry_disagreements", "cull_disagreements", "instance_disagreements")
    sizes = [(160, 90), (333, 211), (640, 360), (97, 61), (1280, 720), (256, 8)]
    tex = rng.integers(0, 256, size=(8, 8, 4), dtype=np.uint8)
    pixels = checks = 0
    per_kind = {"plain": 0, "soups": 0, "lists": 0}
    # The distance cull is the one shortcut that rests on a stated condition instead of a bound (a culled box can hold a triangle
    # that the ray grazes and whose computed distance is wrong by more than the margin): its disagreements are recorded and counted
    # and the run goes on.  Any other disagreement, or a flagged frame that differs from the plain one, ends the run.
    cull_rays, cull_findings = 0, []

    def judge(record, bad, same):
        nonlocal cull_rays
        hard = {n: v for n, v in bad.items() if n != "cull_disagreements"}
        if hard or not same:
            print(json.dumps(record))
            sys.exit(1)
        if bad:
            cull_rays += bad["cull_disagreements"]
            if len(cull_findings) < 20:
                cull_findings.append(record)
            print(json.dumps(record), file=sys.stderr, flush=True)

    def wild(values):
        return tuple(float(rng.choice(special)) if rng.uniform() < a.wild else float(v) for v in values)

    def view(spread):
        eye = rng.uniform(-1.2 * spread - 2, 1.2 * spread + 2, 3)
        at = rng.uniform(-spread, spread, 3)
        if np.linalg.norm(eye - at) < 1e-3:
            at = eye + np.array([0.3, -0.2, 0.9])
        up = (0, 1, 0) if rng.uniform() < 0.8 else tuple(rng.normal(size=3))
        cam = capi.Camera(capi.Vec3(*wild(eye)), capi.Vec3(*wild(at)), capi.Vec3(*wild(up)), wild([rng.uniform(15, 120)])[0])
        if a.wild > 0 and rng.uniform() < 0.5:
            light = capi.Light(capi.Vec3(*wild(rng.normal(size=3))), capi.Vec3(*wild(rng.uniform(-0.5, 2.0, 3))))
        else:
            l = po.lib().orc_make_light(po.vec3(*rng.normal(size=3)), po.vec3(*rng.uniform(0.3, 1.6, 3)))
            light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
        return cam, light

    def describe(cam, light, tfs):
        return {"camera": [repr(v) for v in (cam.position.x, cam.position.y, cam.position.z, cam.look_at.x, cam.look_at.y, cam.look_at.z,
                                             cam.up.x, cam.up.y, cam.up.z, cam.fov)],
                "light": [repr(v) for v in (light.direction.x, light.direction.y, light.direction.z, light.color.x, light.color.y, light.color.z)],
                "placements": [[repr(v) for v in (t.position.x, t.position.y, t.position.z, t.rotation.x, t.rotation.y, t.rotation.z)] for t in tfs]}

    def checked_plain(c, cam, light, tfs, w, h, what, k):
        nonlocal pixels, checks
        a_ = c.render_host(cam, light, tfs, w, h, flags=capi.RT_FLAG_VERIFY_SKIPS)
        cnt = c.last_counters()
        b_ = c.render_host(cam, light, tfs, w, h)
        same = all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(a_, b_))
        bad = {n: cnt[n] for n in kinds if cnt[n]}
        if bad or not same:
            judge({"scene": k, "seed": a.seed, "kind": what, "size": [w, h], "disagreements": bad, "planes_equal": same, **describe(cam, light, tfs)}, bad, same)
        pixels += w * h
        checks += cnt["skip_checks"]

    for k in range(a.scenes):
        kind = k % 4
        if kind in (0, 1):
            spread = float(rng.choice([0.5, 1.5, 5.0, 50.0]))
            tfs = []
            for m in range(3):
                rot = rng.uniform(-3.2, 3.2, 3) if (m < 2 or rng.uniform() < 0.3) else rng.uniform(-0.05, 0.05, 3)
                tfs.append(capi.Transform(capi.Vec3(*wild(rng.uniform(-spread, spread, 3))), capi.Vec3(*wild(rot))))
            cam, light = view(spread)
            w, h = sizes[(k // 2) % len(sizes)]
            checked_plain(ctx, cam, light, tfs, w, h, "plain", k)
            per_kind["plain"] += 1
        elif kind == 2:
            m = int(rng.integers(1, 5))
            with capi.Context(0) as c:
                tfs = []
                for _ in range(m):
                    flat = rng.uniform() < 0.5
                    n = int(rng.choice([40, 200, 800])) if flat else int(rng.choice([1, 5, 64, 300]))
                    soup = (rng.normal(size=(n, 24)) * rng.choice([0.3, 1.0, 3.0])).astype(np.float32)
                    if flat:
                        axis = int(rng.integers(3))
                        soup[:, [axis, 3 + axis, 6 + axis]] = np.float32(rng.normal() * 0.3)
                        soup[:, :9] *= np.float32(rng.choice([1.0, 4.0]))
                        soup[:, [axis, 3 + axis, 6 + axis]] = soup[0, axis]
                    t, nd = po.build_bvh(soup)
                    c.upload_mesh(t, nd, tex)
                    tfs.append(capi.Transform(capi.Vec3(*wild(rng.uniform(-1.5, 1.5, 3))), capi.Vec3(*wild(rng.uniform(-3.2, 3.2, 3)))))
                cam, light = view(2.0)
                checked_plain(c, cam, light, tfs, 200, 120, "soups", k)
            per_kind["soups"] += 1
        else:
            n = int(rng.integers(1, 41))
            spread = float(rng.choice([1.0, 3.0, 8.0]))
            ids = [int(rng.choice([0, 1, 1, 1, 2])) for _ in range(n)]
            tfs = [capi.Transform(capi.Vec3(*wild(rng.uniform(-spread, spread, 3))), capi.Vec3(*wild(rng.uniform(-3.2, 3.2, 3)))) for _ in range(n)]
            if n > 3 and rng.uniform() < 0.3:
                tfs[n - 1] = tfs[0]; ids[n - 1] = ids[0]
            cam, light = view(spread)
            w, h = [(160, 90), (97, 61), (200, 40)][(k // 4) % 3]
            lst = capi.Context.make_instances(ids, tfs)
            planes = []
            for flags in (capi.RT_FLAG_VERIFY_SKIPS, 0):
                rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
                hi = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
                htri = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
                t = torch.empty((h, w), dtype=torch.int32, device="cuda:0")      # the bits of the distance
                sh = torch.empty((h, w), dtype=torch.uint8, device="cuda:0")
                ctx.render_instanced(cam, light, lst, w, h, rgb, hit_instance=hi, hit_tri=htri, hit_t=t, shadow=sh, flags=flags)
                if flags:
                    cnt = ctx.last_counters()
                ctx.sync()
                planes.append((rgb, hi, htri, t, sh))
            same = all(torch.equal(x, y) for x, y in zip(*planes))
            bad = {nm: cnt[nm] for nm in kinds if cnt[nm]}
            if bad or not same:
                judge({"scene": k, "seed": a.seed, "kind": "lists", "size": [w, h], "ids": ids, "disagreements": bad, "planes_equal": same,
                       **describe(cam, light, tfs)}, bad, same)
            pixels += w * h
            checks += cnt["skip_checks"]
            per_kind["lists"] += 1
        if (k + 1) % 100 == 0:
            print(f"[soak] {k + 1} scenes checked themselves: {cull_rays} rays where the distance cull changed the outcome, no other disagreement", file=sys.stderr, flush=True)
    print(json.dumps({"scenes": a.scenes, "seed": a.seed, "mode": "RT_FLAG_VERIFY_SKIPS: shortcuts against the reference's way, on the device, ray by ray",
                      "scenes_by_kind": per_kind, "pixels": pixels, "decisions_examined": checks,
                      "disagreements": {"vis": 0, "shadow_vis": 0, "entry": 0, "instance": 0, "cull": cull_rays},
                      "cull_findings": cull_findings, "planes_equal_to_the_unflagged_frame": True, "wild": a.wild}))
    ctx.close()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scenes", type=int, default=500)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--wild", type=float, default=0.0,
                   help="probability with which each scalar of a scene (placements, camera, field of view, light) is replaced by a special "
                        "value: zeros of either sign, infinities, NaN, 1e30, 1e-30, a denormal, 1e-7, 90/180/360 for angles")
    p.add_argument("--full-size", action="store_true", help="plain scenes at 1920x1080, 1280x720, 3840x2160 and 1917x1083 instead of small frames")
    p.add_argument("--spp", type=int, default=1, help="plain scenes also through rt_render_frame_sampled with this many samples and a second light")
    p.add_argument("--all-paths", action="store_true", help="plain scenes also through rt_render_frames and rt_render_frame_sampled (RGB)")
    p.add_argument("--instanced", action="store_true",
                   help="lists of 1..40 placed meshes through rt_render_frame_instanced against the reference's loop over the list")
    p.add_argument("--mutate", action="store_true", help="with --instanced: each list is the previous one with a few entries changed (kept hierarchy patched)")
    p.add_argument("--rebuilds", action="store_true", help="one context whose second mesh gets a new random soup every step (rebuild / replace routes)")
    p.add_argument("--multi", action="store_true", help="scenes of one to six random soups through the plain, batched and instanced frame paths")
    p.add_argument("--soups", action="store_true", help="random triangle soups with special values among their numbers, both builders and packers")
    p.add_argument("--verify-skips", action="store_true",
                   help="the kernels check their own shortcuts (RT_FLAG_VERIFY_SKIPS) on random plain scenes, random soups and instance lists; no oracle in the loop")
    a = p.parse_args()
    if a.verify_skips:
        return verify_skips(a)
    if a.instanced:
        return instanced(a)
    if a.soups:
        return soups(a)
    if a.multi:
        return multi(a)
    if a.rebuilds:
        return rebuilds(a)
    import torch  # noqa: F401  (its HIP runtime first)
    from oracle import pyoracle as po
    from raytracer_amd import capi
    po.lib()
    meshes = po.default_meshes()
    ctx = capi.Context(0)
    for m in meshes:
        ctx.upload_mesh(m.tris, m.nodes, m.rgba)
    rng = np.random.default_rng(a.seed)
    sizes = [(160, 90), (200, 120), (97, 61), (33, 130), (256, 8)]
    if a.full_size:
        sizes = [(1920, 1080), (1280, 720), (3840, 2160), (1917, 1083)]
    pixels = hits = 0
    special = [0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e30, -1e30, 1e-30, 1e-42, 1e-7, -1e-7, 1.0, 90.0, 180.0, 360.0]

    def wild(values):
        """Each value replaced by a special one with probability --wild."""
        return tuple(float(rng.choice(special)) if rng.uniform() < a.wild else float(v) for v in values)

    for k in range(a.scenes):
        spread = float(rng.choice([0.5, 1.5, 5.0, 50.0]))
        tfs = []
        for m in range(3):
            pos = rng.uniform(-spread, spread, 3)
            rot = rng.uniform(-3.2, 3.2, 3) if (m < 2 or rng.uniform() < 0.3) else rng.uniform(-0.05, 0.05, 3)
            tfs.append(po.Transform(po.vec3(*wild(pos)), po.vec3(*wild(rot))))
        eye = rng.uniform(-1.2 * spread - 2, 1.2 * spread + 2, 3)
        at = rng.uniform(-spread, spread, 3)
        if np.linalg.norm(eye - at) < 1e-3:
            at = eye + np.array([0.3, -0.2, 0.9])
        fov = wild([rng.uniform(15, 120)])[0]
        up = (0, 1, 0) if rng.uniform() < 0.8 else tuple(rng.normal(size=3))
        ocam = po.Camera(po.vec3(*wild(eye)), po.vec3(*wild(at)), po.vec3(*wild(up)), fov)
        if a.wild > 0 and rng.uniform() < 0.5:       # the light as a caller may fill it in: not normalised, any colour
            olight = po.Light(po.vec3(*wild(rng.normal(size=3))), po.vec3(*wild(rng.uniform(-0.5, 2.0, 3))))
        else:
            olight = po.lib().orc_make_light(po.vec3(*rng.normal(size=3)), po.vec3(*rng.uniform(0.3, 1.6, 3)))
        cam = capi.Camera(capi.Vec3(*ocam.position.tup()), capi.Vec3(*ocam.look_at.tup()), capi.Vec3(*ocam.up.tup()), fov)
        light = capi.Light(capi.Vec3(*olight.direction.tup()), capi.Vec3(*olight.color.tup()))
        w, h = sizes[k % len(sizes)]
        want = po.render_frame(meshes, tfs, ocam, olight, w, h)
        ctfs = [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs]
        rgb, hid, ht, sh = ctx.render_host(cam, light, ctfs, w, h)
        same = (np.array_equal(hid, want.hit_id) and np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32))
                and np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb))
        if same and a.all_paths:
            # the throughput kernel (two frames in one launch) and the sampled kernel at one sample and one light
            import torch
            two = torch.empty((2, h, w, 3), dtype=torch.uint8, device="cuda:0")
            smp = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
            ctx.render_frames(cam, light, [ctfs, ctfs], w, h, two)
            ctx.render_sampled(cam, [light], ctfs, w, h, 1, 0, smp)
            ctx.sync()
            same = (np.array_equal(two[0].cpu().numpy(), want.rgb) and np.array_equal(two[1].cpu().numpy(), want.rgb)
                    and np.array_equal(smp.cpu().numpy(), want.rgb))
        if same and a.spp > 1:
            # several samples and a second light against the CPU restatement (unpinned by the reference beyond one sample, one light)
            import torch
            second = po.Light(po.vec3(*wild(rng.normal(size=3))), po.vec3(*wild(rng.uniform(-0.5, 2.0, 3))))
            wants = po.render_frame_sampled(meshes, tfs, ocam, [olight, second], a.spp, k, w, h)
            smp = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
            sid = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
            ctx.set_option("sampled_split", k % 2)
            ctx.render_sampled(cam, [light, capi.Light(capi.Vec3(*second.direction.tup()), capi.Vec3(*second.color.tup()))], ctfs, w, h, a.spp, k, smp, hit_id=sid)
            ctx.sync()
            same = np.array_equal(smp.cpu().numpy(), wants.rgb) and np.array_equal(sid.cpu().numpy(), wants.hit_id)
        if not same:
            print(json.dumps({"scene": k, "seed": a.seed, "size": [w, h], "ids_differ": int((hid != want.hit_id).sum()),
                              "t_differ": int((ht.view(np.uint32) != want.hit_t.view(np.uint32)).sum()),
                              "shadow_differ": int((sh != want.shadow).sum()), "rgb_differ": int((rgb != want.rgb).any(-1).sum()),
                              "placements": [[repr(v) for v in t.position.tup() + t.rotation.tup()] for t in tfs],
                              "camera": [repr(v) for v in ocam.position.tup() + ocam.look_at.tup() + ocam.up.tup()] + [repr(fov)],
                              "light": [repr(v) for v in olight.direction.tup() + olight.color.tup()]}))
            sys.exit(1)
        pixels += w * h
        hits += int((want.hit_id >= 0).sum())
        if (k + 1) % 100 == 0:
            print(f"[soak] {k + 1} scenes identical", file=sys.stderr, flush=True)
    out = {"scenes": a.scenes, "seed": a.seed, "pixels": pixels, "pixels_with_a_hit": hits,
           "planes": "rgb, hit id, closest t (bits), shadow", "identical": True}
    if a.wild > 0:
        out["wild"] = a.wild
        out["special_values"] = "each scalar of placements, camera, field of view and light replaced with that probability by one of " + repr(special)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
