"""Pins the CPU oracle (oracle/rt_oracle.c) to the golden vectors generated from the compiled
reference (tests/golden/make_golden.py), bit for bit, and -- where oracle/_ref/libref.so is present --
to the compiled reference itself on fresh inputs.  CPU only."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import bits

NAMES = ("drone", "treasure", "ground")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ----------------------------------------------------------------------------------- assets, BVH

def test_textures_decode_to_the_reference_bytes(po, golden_hashes):
    for n in NAMES:
        t = po.decode_webp(os.path.join(po.ASSETS, n + ".webp"))
        want = golden_hashes["textures"][n]
        assert list(t.shape) == want["shape"]
        assert sha(t) == want["sha256"]


@pytest.mark.parametrize("name", NAMES)
def test_bvh_build_matches_reference_permutation_and_nodes(po, golden_bvh, golden_hashes, name):
    """create_bvh (accel/bvh.c:4-67): same in-place triangle permutation, same node array."""
    soup = po.load_obj(os.path.join(po.ASSETS, name + ".obj"))
    assert soup.shape[0] == golden_hashes["meshes"][name]["triangles"]
    post, nodes = po.build_bvh(soup)
    assert np.array_equal(post.view(np.uint32), soup[golden_bvh[name + "_perm"]].view(np.uint32))
    assert sha(post) == golden_hashes["meshes"][name]["triangles_post_sha256"]
    want = golden_bvh[name + "_nodes"]
    assert nodes.shape == want.shape
    for f in ("left", "right", "start", "count"):
        assert np.array_equal(nodes[f], want[f]), f
    for f in ("bmin", "bmax"):
        assert np.array_equal(bits(nodes[f]), bits(want[f])), f


def test_bvh_build_edge_cases(po):
    rng = np.random.default_rng(3)
    # 1..4 triangles: a single leaf
    for n in (1, 2, 4):
        t, nodes = po.build_bvh(rng.normal(size=(n, 24)).astype(np.float32))
        assert nodes.shape[0] == 1 and nodes["count"][0] == n and nodes["left"][0] == -1
    # all centroids equal: no triangle is below the mean, the partition is empty and the node stays a
    # leaf with more than 4 triangles (bvh.c:51-55)
    one = rng.normal(size=(1, 24)).astype(np.float32)
    t, nodes = po.build_bvh(np.repeat(one, 9, axis=0))
    assert nodes.shape[0] == 1 and nodes["count"][0] == 9 and nodes["left"][0] == -1
    # every triangle appears exactly once in the leaves, in order
    t, nodes = po.build_bvh(rng.normal(size=(257, 24)).astype(np.float32))
    leaves = nodes[nodes["left"] < 0]
    order = np.argsort(leaves["start"])
    starts, counts = leaves["start"][order], leaves["count"][order]
    assert starts[0] == 0 and np.array_equal(starts[1:], np.cumsum(counts)[:-1]) and counts.sum() == 257


def test_light_direction_is_the_reference_normalisation(po, golden_hashes):
    got = [int(np.float32(v).view(np.uint32)) for v in po.default_light().direction.tup()]
    assert got == golden_hashes["light_direction_bits"]


# ----------------------------------------------------------------------------------- unit KATs

def test_kat_ray_aabb(po, golden_kat):
    g = golden_kat
    L = po.lib()
    got = np.array([L.orc_ray_aabb(po.Ray(po.vec3(*g["aabb_o"][k]), po.vec3(*g["aabb_d"][k])),
                                   po.Aabb(po.vec3(*g["aabb_lo"][k]), po.vec3(*g["aabb_hi"][k])))
                    for k in range(g["aabb_hit"].shape[0])], np.int32)
    assert np.array_equal(got, g["aabb_hit"])
    assert 0 < got.sum() < got.size               # the vector exercises both outcomes


def test_kat_ray_triangle(po, golden_kat):
    g = golden_kat
    L = po.lib()
    n = g["tri_hit"].shape[0]
    hit = np.zeros(n, np.int32)
    tuv = np.zeros((n, 3), np.float32)
    for k in range(n):
        t, u, v = C.c_float(-1), C.c_float(-1), C.c_float(-1)
        a, b, c = (po.vec3(*g["tri_v"][k, j]) for j in range(3))
        hit[k] = L.orc_ray_triangle(po.Ray(po.vec3(*g["tri_o"][k]), po.vec3(*g["tri_d"][k])), a, b, c,
                                    C.byref(t), C.byref(u), C.byref(v))
        if hit[k]:
            tuv[k] = (t.value, u.value, v.value)
    assert np.array_equal(hit, g["tri_hit"])
    assert np.array_equal(bits(tuv), bits(g["tri_tuv"]))
    assert 0 < hit.sum() < n


def test_kat_transform_ray_and_normal(po, golden_kat):
    g = golden_kat
    L = po.lib()
    n = g["tf_pos"].shape[0]
    rays = np.zeros((n, 6), np.float32)
    nrm = np.zeros((n, 3), np.float32)
    for k in range(n):
        tf = po.Transform(po.vec3(*g["tf_pos"][k]), po.vec3(*g["tf_rot"][k]))
        r = L.orc_transform_ray(po.Ray(po.vec3(*g["tf_o"][k]), po.vec3(*g["tf_d"][k])), tf)
        rays[k] = r.origin.tup() + r.direction.tup()
        nrm[k] = L.orc_transform_normal(po.vec3(*g["tf_d"][k]), tf).tup()
    assert np.array_equal(bits(rays), bits(g["tf_ray"]))
    assert np.array_equal(bits(nrm), bits(g["tf_normal"]))


def test_kat_camera_ray(po, golden_kat):
    g = golden_kat
    L = po.lib()
    n = g["cam_params"].shape[0]
    out = np.zeros((n, 6), np.float32)
    for k in range(n):
        p = g["cam_params"][k]
        cam = po.Camera(po.vec3(*p[0:3]), po.vec3(*p[3:6]), po.vec3(*p[6:9]), float(p[9]))
        r = L.orc_camera_ray(C.byref(cam), float(g["cam_xy"][k, 0]), float(g["cam_xy"][k, 1]), float(g["cam_aspect"][k]))
        out[k] = r.origin.tup() + r.direction.tup()
    assert np.array_equal(bits(out), bits(g["cam_ray"]))


@pytest.mark.parametrize("mesh", range(3))
def test_kat_intersect_bvh(po, meshes, golden_kat, mesh):
    """intersect_bvh (accel/bvh.c:69-109) on the reference-built trees: same hit, t, u, v and triangle."""
    g = golden_kat
    name = NAMES[mesh]
    m = meshes[mesh]
    L = po.lib()
    n = g[f"bvh_{name}_hit"].shape[0]
    hit = np.zeros(n, np.int32)
    tuv = np.zeros((n, 3), np.float32)
    idx = np.full(n, -1, np.int32)
    for k in range(n):
        t, u, v, i = C.c_float(1e30), C.c_float(0), C.c_float(0), C.c_int32(-1)
        ray = po.Ray(po.vec3(*g[f"bvh_{name}_o"][k]), po.vec3(*g[f"bvh_{name}_d"][k]))
        hit[k] = L.orc_intersect_bvh(m.nodes.ctypes.data, 0, ray, m.tris.ctypes.data, C.byref(t), C.byref(u), C.byref(v), C.byref(i))
        if hit[k]:
            tuv[k] = (t.value, u.value, v.value)
            idx[k] = i.value
    assert np.array_equal(hit, g[f"bvh_{name}_hit"])
    assert np.array_equal(idx, g[f"bvh_{name}_idx"])
    assert np.array_equal(bits(tuv), bits(g[f"bvh_{name}_tuv"]))
    assert hit.sum() > n // 10


def test_kat_bicubic(po, golden_kat, golden_frames):
    g = golden_kat
    frame = np.ascontiguousarray(golden_frames["f0_rgb"])
    L = po.lib()
    got = np.array([L.orc_bicubic(frame.ctypes.data, float(g["bic_x"][k]), float(g["bic_y"][k]), 240, 135)
                    for k in range(g["bic_x"].shape[0])], np.uint32)
    assert np.array_equal(got, g["bic_out"])


# ----------------------------------------------------------------------------------- whole frames

@pytest.mark.parametrize("frame", [0, 45, 95])
def test_small_frames_every_plane(po, meshes, golden_frames, frame):
    fr = po.render_frame(meshes, po.animation_transforms(frame), po.default_camera(), po.default_light(), 240, 135)
    g = golden_frames
    assert np.array_equal(fr.hit_id, g[f"f{frame}_hit_id"])
    assert np.array_equal(bits(fr.hit_t), bits(g[f"f{frame}_hit_t"]))
    assert np.array_equal(fr.shadow, g[f"f{frame}_shadow"])
    assert np.array_equal(fr.rgb, g[f"f{frame}_rgb"])


@pytest.mark.parametrize("key", ["720x540_f0", "720x540_f45", "720x540_f95", "1920x1080_f0", "1920x1080_f45"])
def test_full_size_hashes(po, meshes, golden_hashes, key):
    size, f = key.split("_f")
    w, h = (int(v) for v in size.split("x"))
    fr = po.render_frame(meshes, po.animation_transforms(int(f)), po.default_camera(), po.default_light(), w, h)
    want = golden_hashes["frames"][key]
    assert int((fr.hit_id >= 0).sum()) == want["hit_pixels"]
    assert int((fr.shadow == 1).sum()) == want["shadowed_pixels"]
    assert sha(fr.hit_id) == want["hit_id"] and sha(fr.hit_t) == want["hit_t"]
    assert sha(fr.shadow) == want["shadow"] and sha(fr.rgb) == want["rgb"]
    ppm = hashlib.sha1(b"P6\n%d %d\n255\n" % (w, h) + fr.rgb.tobytes()).hexdigest()
    assert ppm == want["ppm_sha1"]


def test_reference_driver_frame0_ppm_hash(golden_hashes):
    """Frame 0 of the reference's own `make run` configuration (800x600 at scale 0.9 = 720x540) written as a
    binary PPM: SHA-1 measured on the full reference program built with -ffp-contract=off."""
    assert golden_hashes["frames"]["720x540_f0"]["ppm_sha1"] == "e1f14d00082a7e6d4ba2940e3c19cd73d25ff750"


def test_frame_is_deterministic_across_thread_counts(po, meshes):
    tfs = po.animation_transforms(30)
    a = po.render_frame(meshes, tfs, po.default_camera(), po.default_light(), 160, 90, threads=1)
    b = po.render_frame(meshes, tfs, po.default_camera(), po.default_light(), 160, 90, threads=4)
    assert np.array_equal(a.rgb, b.rgb) and np.array_equal(a.hit_id, b.hit_id) and a.counters == b.counters


# ----------------------------------------------------------------------------------- compiled reference

needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref.so")),
                               reason="oracle/_ref/libref.so is built only where the reference sources exist")


@needs_ref
def test_oracle_equals_compiled_reference_on_fresh_placements(po, meshes):
    """Random placements, cameras and lights that no fixture holds."""
    soups = [po.load_obj(os.path.join(po.ASSETS, n + ".obj")) for n in NAMES]
    scene = po.RefScene(soups, [m.rgba for m in meshes])
    rng = np.random.default_rng(11)
    try:
        for trial in range(5):
            tfs = [po.Transform(po.vec3(*rng.uniform(-1.5, 1.5, 3)), po.vec3(*rng.uniform(-3.2, 3.2, 3))) for _ in range(2)]
            tfs.append(po.Transform(po.vec3(0, 0, 0), po.vec3(*rng.uniform(-0.05, 0.05, 3))))
            eye = rng.uniform(-4, 4, 3)
            eye[1] = abs(eye[1]) + 1.0
            cam = po.Camera(po.vec3(*eye), po.vec3(0, 0.3, 0), po.vec3(0, 1, 0), float(rng.uniform(30, 90)))
            light = po.Light()
            d, c = po.vec3(*rng.uniform(-1, 1, 3)), po.vec3(1.1, 0.9, 1.3)
            po.ref().refc_make_light(C.byref(d), C.byref(c), C.byref(light))
            olight = po.lib().orc_make_light(d, c)
            assert bits(np.array(light.direction.tup())).tolist() == bits(np.array(olight.direction.tup())).tolist()
            want = scene.render(tfs, cam, light, 192, 108)
            got = po.render_frame(meshes, tfs, cam, olight, 192, 108)
            assert np.array_equal(got.hit_id, want.hit_id)
            assert np.array_equal(bits(got.hit_t), bits(want.hit_t))
            assert np.array_equal(got.shadow, want.shadow)
            assert np.array_equal(got.rgb, want.rgb)
    finally:
        scene.close()


@pytest.mark.parametrize("frame", [7, 23, 60, 88])
def test_oracle_matches_animation_hashes(po, meshes, frame):
    """Frames of the 96-frame run other than the three that have full golden planes."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    gold = json.load(open(os.path.join(GOLDEN, "animation_hashes.json")))
    fr = po.render_frame(meshes, po.animation_transforms(frame), po.default_camera(), po.default_light(), gold["width"], gold["height"], debug=False)
    assert hashlib.sha256(fr.rgb.tobytes()).hexdigest() == gold["frames"][frame]
