#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the COMPILED REFERENCE.

Run from the repo root, in a container that has /root/reference:

    make oracle && python tests/golden/make_golden.py

Everything written here is data (inputs and the reference's outputs); no reference source
is copied.  The producer is oracle/_ref/libref.so = the reference's own translation units
math/vec3.c math/mat4.c math/ray.c geometry/aabb.c accel/bvh.c render/camera.c render/light.c
utils/image.c compiled as they are, driven by oracle/ref_compose.c (the reference's scene.c and
geometry/mesh.c need <webp/...> headers that this image lacks, so the pixel loop around the
reference calls and the texel fetch are the composition driver's).

Files:
  frames_240x135.npz   per-pixel rgb / hit_id / hit_t / shadow, frames 0, 45, 95
  hashes.json          SHA-256 of the same planes at 720x540 (frames 0,45,95) and 1920x1080 (0,45),
                       SHA-256 of the decoded textures, triangle/node counts
  bvh.npz              per mesh: permutation file order -> post-build order, flat node array
  kat.npz              known-answer vectors for the unit functions (random + adversarial inputs)
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NAMES = ("drone", "treasure", "ground")


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    R = po.ref()
    soups = [po.load_obj(os.path.join(po.ASSETS, n + ".obj")) for n in NAMES]
    texs = [po.decode_webp(os.path.join(po.ASSETS, n + ".webp")) for n in NAMES]
    scene = po.RefScene(soups, texs)
    cam = po.default_camera()
    light = po.Light()
    d, c = po.vec3(1, 1, -1), po.vec3(1.4, 1.4, 1.4)
    R.refc_make_light(C.byref(d), C.byref(c), C.byref(light))

    # ---- BVH: permutation + nodes
    bvh = {}
    for i, n in enumerate(NAMES):
        post = scene.triangles(i)
        index = {}
        for k, row in enumerate(soups[i]):
            index.setdefault(row.tobytes(), []).append(k)
        perm = np.array([index[row.tobytes()].pop(0) for row in post], dtype=np.int32)
        assert np.array_equal(soups[i][perm].view(np.uint32), post.view(np.uint32))
        bvh[n + "_perm"] = perm
        bvh[n + "_nodes"] = scene.nodes(i)
    np.savez_compressed(os.path.join(OUT, "bvh.npz"), **bvh)

    # ---- small frames, every plane
    small = {}
    for f in (0, 45, 95):
        fr = scene.render(po.animation_transforms(f), cam, light, 240, 135)
        small[f"f{f}_rgb"] = fr.rgb
        small[f"f{f}_hit_id"] = fr.hit_id
        small[f"f{f}_hit_t"] = fr.hit_t
        small[f"f{f}_shadow"] = fr.shadow
    np.savez_compressed(os.path.join(OUT, "frames_240x135.npz"), **small)

    # ---- hashes at the real sizes
    hashes = {"textures": {n: {"sha256": sha(t), "shape": list(t.shape)} for n, t in zip(NAMES, texs)},
              "meshes": {n: {"triangles": int(s.shape[0]), "nodes": int(bvh[n + "_nodes"].shape[0]),
                             "triangles_post_sha256": sha(scene.triangles(i))}
                         for i, (n, s) in enumerate(zip(NAMES, soups))},
              "light_direction_bits": [int(np.float32(v).view(np.uint32)) for v in light.direction.tup()],
              "frames": {}}
    for (w, h), frames in (((720, 540), (0, 45, 95)), ((1920, 1080), (0, 45))):
        for f in frames:
            fr = scene.render(po.animation_transforms(f), cam, light, w, h)
            hashes["frames"][f"{w}x{h}_f{f}"] = {
                "rgb": sha(fr.rgb), "hit_id": sha(fr.hit_id), "hit_t": sha(fr.hit_t), "shadow": sha(fr.shadow),
                "hit_pixels": int((fr.hit_id >= 0).sum()), "shadowed_pixels": int((fr.shadow == 1).sum()),
                "ppm_sha1": hashlib.sha1(b"P6\n%d %d\n255\n" % (w, h) + fr.rgb.tobytes()).hexdigest(),
            }
    json.dump(hashes, open(os.path.join(OUT, "hashes.json"), "w"), indent=1, sort_keys=True)

    # ---- unit known-answer vectors
    rng = np.random.default_rng(20251111)
    N = 4096
    kat = {}

    def rand_dirs(n):
        v = rng.normal(size=(n, 3)).astype(np.float32)
        v[: n // 8, 0] = 0.0                      # axis-parallel components: 1/0 = inf in the slab test
        v[n // 16: n // 8, 1] = 0.0
        v[n // 8: n // 6, 2] = -0.0
        return v

    # ray_aabb_intersect
    o = (rng.uniform(-4, 4, size=(N, 3))).astype(np.float32)
    dd = rand_dirs(N)
    lo = rng.uniform(-3, 3, size=(N, 3)).astype(np.float32)
    hi = lo + rng.uniform(0, 3, size=(N, 3)).astype(np.float32)
    hi[: N // 4, 1] = lo[: N // 4, 1]              # zero-thickness boxes like the ground's
    o[N // 4: N // 3] = ((lo + hi) * 0.5)[N // 4: N // 3]   # rays that start inside
    o[N // 3: N // 3 + 200, 1] = lo[N // 3: N // 3 + 200, 1]  # origin on a box face with...
    dd[N // 3: N // 3 + 100, 1] = 0.0                          # ...a zero direction component: 0*inf
    res = np.zeros(N, np.int32)
    for k in range(N):
        ray = po.Ray(po.vec3(*o[k]), po.vec3(*dd[k]))
        box = po.Aabb(po.vec3(*lo[k]), po.vec3(*hi[k]))
        res[k] = R.refc_ray_aabb(C.byref(ray), C.byref(box))
    kat.update(aabb_o=o, aabb_d=dd, aabb_lo=lo, aabb_hi=hi, aabb_hit=res)

    # ray_triangle_intersect
    to = rng.uniform(-2, 2, size=(N, 3)).astype(np.float32)
    tgt = rng.uniform(-1, 1, size=(N, 3)).astype(np.float32)
    tv = rng.uniform(-1.5, 1.5, size=(N, 3, 3)).astype(np.float32)
    td = tgt - to
    td[: N // 2] = (tv[: N // 2].mean(axis=1) - to[: N // 2])      # half aimed at the triangle
    k0 = N // 2
    td[k0: k0 + 300] = (tv[k0: k0 + 300, 0] * 0.5 + tv[k0: k0 + 300, 1] * 0.5 - to[k0: k0 + 300])   # aimed at an edge
    td[k0 + 300: k0 + 400] = (tv[k0 + 300: k0 + 400, 0] - to[k0 + 300: k0 + 400])                    # aimed at a vertex
    tv[k0 + 400: k0 + 450, 2] = tv[k0 + 400: k0 + 450, 1]                                           # degenerate triangles
    td = (td / np.maximum(np.linalg.norm(td, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    hit = np.zeros(N, np.int32)
    tuv = np.zeros((N, 3), np.float32)
    for k in range(N):
        ray = po.Ray(po.vec3(*to[k]), po.vec3(*td[k]))
        a, b, c = (po.vec3(*tv[k, j]) for j in range(3))
        t, u, v = C.c_float(-1), C.c_float(-1), C.c_float(-1)
        hit[k] = R.refc_ray_triangle(C.byref(ray), C.byref(a), C.byref(b), C.byref(c), C.byref(t), C.byref(u), C.byref(v))
        tuv[k] = (t.value, u.value, v.value) if hit[k] else (0, 0, 0)
    kat.update(tri_o=to, tri_d=td, tri_v=tv, tri_hit=hit, tri_tuv=tuv)

    # transform_ray / transform_normal
    M = 1024
    pos = rng.uniform(-3, 3, size=(M, 3)).astype(np.float32)
    rot = rng.uniform(-4, 4, size=(M, 3)).astype(np.float32)
    rot[:64] = 0.0
    pos[:32] = 0.0
    ro = rng.uniform(-5, 5, size=(M, 3)).astype(np.float32)
    rd = rand_dirs(M)
    rd = (rd / np.maximum(np.linalg.norm(rd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    out_ray = np.zeros((M, 6), np.float32)
    out_nrm = np.zeros((M, 3), np.float32)
    for k in range(M):
        tf = po.Transform(po.vec3(*pos[k]), po.vec3(*rot[k]))
        ray = po.Ray(po.vec3(*ro[k]), po.vec3(*rd[k]))
        r2 = po.Ray()
        R.refc_transform_ray(C.byref(ray), C.byref(tf), C.byref(r2))
        out_ray[k] = r2.origin.tup() + r2.direction.tup()
        n = po.vec3(*rd[k])
        n2 = po.Vec3()
        R.refc_transform_normal(C.byref(n), C.byref(tf), C.byref(n2))
        out_nrm[k] = n2.tup()
    kat.update(tf_pos=pos, tf_rot=rot, tf_o=ro, tf_d=rd, tf_ray=out_ray, tf_normal=out_nrm)

    # get_camera_ray
    cams = np.zeros((M, 10), np.float32)
    cams[:, 0:3] = rng.uniform(-5, 5, size=(M, 3))
    cams[:, 3:6] = rng.uniform(-1, 1, size=(M, 3))
    cams[:, 6:9] = (0, 1, 0)
    cams[:, 9] = rng.uniform(20, 100, size=M)
    cams[0] = (-3, 3, -3, 0, 0, 0, 0, 1, 0, 60)
    xy = rng.uniform(0, 1, size=(M, 2)).astype(np.float32)
    asp = rng.uniform(0.5, 2.5, size=M).astype(np.float32)
    cray = np.zeros((M, 6), np.float32)
    for k in range(M):
        cm = po.Camera(po.vec3(*cams[k, 0:3]), po.vec3(*cams[k, 3:6]), po.vec3(*cams[k, 6:9]), float(cams[k, 9]))
        r2 = po.Ray()
        R.refc_camera_ray(C.byref(cm), float(xy[k, 0]), float(xy[k, 1]), float(asp[k]), C.byref(r2))
        cray[k] = r2.origin.tup() + r2.direction.tup()
    kat.update(cam_params=cams, cam_xy=xy, cam_aspect=asp, cam_ray=cray)

    # intersect_bvh on the three reference-built trees (mesh space rays)
    for i, n in enumerate(NAMES):
        nodes = bvh[n + "_nodes"]
        lo_, hi_ = nodes["bmin"][0], nodes["bmax"][0]
        K = 2048
        bo = rng.uniform(lo_ - 2.0, hi_ + 2.0, size=(K, 3)).astype(np.float32)
        bo[:, 1] = np.abs(bo[:, 1]) + hi_[1] + 0.1
        aim = rng.uniform(lo_, hi_, size=(K, 3)).astype(np.float32)
        bd = aim - bo
        bd = (bd / np.maximum(np.linalg.norm(bd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
        bh = np.zeros(K, np.int32)
        bres = np.zeros((K, 3), np.float32)
        bidx = np.full(K, -1, np.int32)
        for k in range(K):
            ray = po.Ray(po.vec3(*bo[k]), po.vec3(*bd[k]))
            t, u, v, idx = C.c_float(1e30), C.c_float(0), C.c_float(0), C.c_int32(-1)
            bh[k] = R.refc_intersect(scene.handles[i], C.byref(ray), C.byref(t), C.byref(u), C.byref(v), C.byref(idx))
            if bh[k]:
                bres[k] = (t.value, u.value, v.value)
                bidx[k] = idx.value
        kat.update({f"bvh_{n}_o": bo, f"bvh_{n}_d": bd, f"bvh_{n}_hit": bh, f"bvh_{n}_tuv": bres, f"bvh_{n}_idx": bidx})

    # bicubic_interpolate on frame 0 of the small fixture
    fr0 = small["f0_rgb"]
    bx = rng.uniform(-0.5, 240.5, size=M).astype(np.float32)
    by = rng.uniform(-0.5, 135.5, size=M).astype(np.float32)
    bx[:16] = np.arange(16, dtype=np.float32) * 15.0
    by[:16] = np.arange(16, dtype=np.float32) * 8.0
    bout = np.zeros(M, np.uint32)
    fr0c = np.ascontiguousarray(fr0)
    for k in range(M):
        bout[k] = R.refc_bicubic(fr0c.ctypes.data, float(bx[k]), float(by[k]), 240, 135)
    kat.update(bic_x=bx, bic_y=by, bic_out=bout)

    np.savez_compressed(os.path.join(OUT, "kat.npz"), **kat)
    scene.close()
    for fn in ("bvh.npz", "frames_240x135.npz", "hashes.json", "kat.npz"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


if __name__ == "__main__":
    main()
