"""rt_bvh_build_device: the reference's BVH build (accel/bvh.c:4-59) run level by level on the GPU must
leave the same bits as the host build -- triangle order and node array -- which in turn is pinned to
the compiled reference (test_abi_and_host.py, golden BVH arrays)."""
import os

import numpy as np
import pytest

from test_abi_and_host import signed_zero_soup

pytestmark = pytest.mark.gpu

NAMES = ("drone", "treasure", "ground")


@pytest.fixture(scope="module")
def capi():
    from raytracer_amd import capi as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


def same(capi, ctx, soup):
    want_t, want_n = capi.build_bvh(soup)
    got_t, got_n = ctx.build_bvh(soup)
    assert got_n.shape == want_n.shape
    assert np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32))
    assert got_n.tobytes() == want_n.tobytes()
    return want_n


def soup_of(n, seed, spread="uniform"):
    rng = np.random.default_rng(seed)
    t = np.zeros((n, 24), np.float32)
    if spread == "uniform":
        c = rng.random((n, 1, 3), dtype=np.float32) * 10
    elif spread == "skewed":                       # means far from medians: one-sided splits, deep trees
        c = (rng.random((n, 1, 3)) ** 8 * 100).astype(np.float32)
    elif spread == "outlier":                      # one far triangle drags the mean: all but one key smaller
        c = rng.random((n, 1, 3), dtype=np.float32)
        c[n // 2] = 1.0e4
    elif spread == "grid":                         # many equal keys
        c = rng.integers(0, 4, size=(n, 1, 3)).astype(np.float32)
    else:
        raise ValueError(spread)
    t[:, :9] = (c + rng.random((n, 3, 3), dtype=np.float32) * 0.1).reshape(n, 9)
    t[:, 9:] = rng.random((n, 15), dtype=np.float32)
    return t


@pytest.mark.parametrize("name", NAMES)
def test_device_build_equals_host_build_on_the_assets(capi, ctx, po, golden_bvh, name):
    soup = po.load_obj(os.path.join(po.ASSETS, name + ".obj"))
    nodes = same(capi, ctx, soup)
    assert np.array_equal(nodes["left"], golden_bvh[name + "_nodes"]["left"])
    st = ctx.build_stats()
    assert st["levels"] >= 2 and st["launches"] > 0


@pytest.mark.parametrize("n", [1, 2, 4, 5, 6, 7, 9, 63, 64, 65, 129, 1000, 4097])
def test_device_build_small_and_ragged_sizes(capi, ctx, n):
    same(capi, ctx, soup_of(n, n))


@pytest.mark.parametrize("spread", ["skewed", "outlier", "grid"])
def test_device_build_one_sided_and_tied_splits(capi, ctx, spread):
    for n in (37, 700, 20000):
        same(capi, ctx, soup_of(n, 11, spread))


def test_device_build_sorted_keys(capi, ctx):
    """Ascending keys: all smaller keys first (no rotation); descending: every smaller key rotates the queue."""
    n = 3000
    t = soup_of(n, 3)
    order = np.argsort(t[:, 0])
    same(capi, ctx, t[order])
    same(capi, ctx, t[order[::-1]])


def test_device_build_failed_partitions_and_signed_zeros(capi, ctx):
    one = soup_of(1, 7)
    same(capi, ctx, np.repeat(one, 9, axis=0))            # nine identical triangles stay one leaf (bvh.c:51-55)
    same(capi, ctx, np.repeat(one, 300, axis=0))
    mixed = np.concatenate([np.repeat(one, 40, axis=0), soup_of(200, 8)])
    same(capi, ctx, mixed)
    same(capi, ctx, signed_zero_soup(300, 5))
    same(capi, ctx, signed_zero_soup(5000, 6))


def test_device_build_on_nan_infinite_and_overflowing_vertices(capi, ctx):
    """tests/adversarial.py::nonfinite_soups: bounds that are NaN or lie beyond the +-1e30 an empty box starts with take no part
    in a node's box (aabb.c:3-23), a NaN centroid fails every comparison of the partition -- same bits as the host build."""
    import adversarial
    n = 0
    for name, soup in adversarial.nonfinite_soups():
        same(capi, ctx, soup)
        n += 1
    assert n >= 40


def test_upload_soup_of_nan_and_infinite_vertices_renders_like_the_reference(capi, po):
    """The same soups through rt_mesh_upload_soup with build and packing on the device (device_build_min = 1): post-build
    triangle order equal to the oracle's build, frames equal to the oracle's reference loop over that build."""
    import adversarial
    tex = np.full((4, 4, 4), 180, np.uint8)
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    pl = po.default_light()
    light = capi.Light(capi.Vec3(*pl.direction.tup()), capi.Vec3(*pl.color.tup()))
    ident = [po.Transform(po.vec3(0, 0, 0), po.vec3(0.2, 0.4, 0.1))]
    ctf = [capi.Transform(capi.Vec3(0, 0, 0), capi.Vec3(0.2, 0.4, 0.1))]
    n = 0
    for name, soup in adversarial.nonfinite_soups(sizes=(60, 500)):
        want_t, want_n = po.build_bvh(soup)
        want = po.render_frame([po.HostMesh(want_t, want_n, tex)], ident, po.default_camera(), pl, 96, 54)
        with capi.Context(0) as c:
            c.set_option("device_build_min", 1)
            mid, post = c.upload_soup(soup, tex, want_triangles=True)
            assert np.array_equal(post.view(np.uint32), want_t.view(np.uint32)), name
            rgb, hid, ht, sh = c.render_host(cam, light, ctf, 96, 54)
            assert np.array_equal(hid, want.hit_id) and np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32)), name
            assert np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb), name
        n += 1
    assert n >= 20


def test_device_build_empty_and_argument_errors(capi, ctx):
    import ctypes as C
    L = capi.lib()
    used = C.c_int32(0)
    nodes = np.zeros(2, capi.NODE_DTYPE)
    assert L.rt_bvh_build_device(ctx._h, None, 0, nodes.ctypes.data, 2, C.byref(used)) == 0 and used.value == 1
    tris = soup_of(100, 1)
    assert L.rt_bvh_build_device(ctx._h, tris.ctypes.data, 100, nodes.ctypes.data, 2, C.byref(used)) == capi.RT_ERR_CAPACITY
    assert L.rt_bvh_build_device(ctx._h, None, 100, nodes.ctypes.data, 2, C.byref(used)) == capi.RT_ERR_INVALID
    assert L.rt_bvh_build_device(None, tris.ctypes.data, 100, nodes.ctypes.data, 2, C.byref(used)) == capi.RT_ERR_INVALID


def test_device_build_large_mesh_and_render(capi, ctx, po):
    """Half a million triangles: same bits as the host build, and the uploaded result renders like the host-built one."""
    import time
    soup = soup_of(500_000, 21)
    t0 = time.perf_counter()
    want_t, want_n = capi.build_bvh(soup)
    t1 = time.perf_counter()
    got_t, got_n = ctx.build_bvh(soup)
    t2 = time.perf_counter()
    assert np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32)) and got_n.tobytes() == want_n.tobytes()
    print(f"\n500k triangles: host build {1e3 * (t1 - t0):.0f} ms, device build {1e3 * (t2 - t1):.0f} ms "
          f"(kernels {ctx.build_stats()['device_ms']:.1f} ms, {ctx.build_stats()['launches']} launches)")


def bake(tris, angle, offset):
    """Vertices and normals of a triangle soup rotated about y and moved, in float32 (a mesh that deforms on the
    host; any float recipe will do, GPU and oracle get the same numbers)."""
    c, s = np.float32(np.cos(angle)), np.float32(np.sin(angle))
    rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    out = tris.copy()
    for k in range(3):
        out[:, 3 * k:3 * k + 3] = (tris[:, 3 * k:3 * k + 3] @ rot.T + np.asarray(offset, np.float32)).astype(np.float32)
        out[:, 15 + 3 * k:18 + 3 * k] = (tris[:, 15 + 3 * k:18 + 3 * k] @ rot.T).astype(np.float32)
    return out


def test_mesh_rebuilt_every_frame_renders_like_the_reference_loop(capi, ctx, po):
    """The moving mesh's vertices are rewritten each frame, its BVH is rebuilt (on the GPU and on the host) and
    swapped in with rt_mesh_replace; frames equal the reference loop's on the same triangles."""
    import torch
    meshes = po.default_meshes()
    soup = po.load_obj(os.path.join(po.ASSETS, "drone.obj"))
    c = capi.Context(0)
    for m in meshes:
        c.upload_mesh(m.tris, m.nodes, m.rgba)
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    pl = po.default_light()
    light = capi.Light(capi.Vec3(*pl.direction.tup()), capi.Vec3(*pl.color.tup()))
    w, h = 160, 90
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    hid = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
    for frame, builder in ((0, c.build_bvh), (17, capi.build_bvh), (40, c.build_bvh)):
        baked = bake(soup, 0.05 * frame, (2.0 * np.cos(0.05 * frame), 1.0, 2.0 * np.sin(0.05 * frame)))
        tris, nodes = builder(baked)
        c.replace_mesh(0, tris, nodes)
        t1 = po.animation_transforms(frame)[1]
        tfs = [capi.Transform(), capi.Transform(capi.Vec3(*t1.position.tup()), capi.Vec3(*t1.rotation.tup())), capi.Transform()]
        c.render(cam, light, tfs, w, h, rgb, hit_id=hid)
        c.sync()
        want_t, want_n = po.build_bvh(baked)
        ref_meshes = [po.HostMesh(want_t, want_n, meshes[0].rgba), meshes[1], meshes[2]]
        ptfs = [po.Transform(), po.animation_transforms(frame)[1], po.Transform()]
        want = po.render_frame(ref_meshes, ptfs, po.default_camera(), po.default_light(), w, h)
        assert np.array_equal(hid.cpu().numpy(), want.hit_id)
        assert np.array_equal(rgb.cpu().numpy(), want.rgb)
        assert (want.hit_id >> 24 == 0).any()                      # the rebuilt mesh is in view
    # a different triangle count reallocates
    tris, nodes = capi.build_bvh(bake(soup[:100], 0.3, (0, 1, 0)))
    c.replace_mesh(0, tris, nodes)
    c.render(cam, light, tfs, w, h, rgb, hit_id=hid)
    c.sync()
    want = po.render_frame([po.HostMesh(tris, nodes, meshes[0].rgba), meshes[1], meshes[2]], ptfs, po.default_camera(), po.default_light(), w, h)
    assert np.array_equal(hid.cpu().numpy(), want.hit_id) and np.array_equal(rgb.cpu().numpy(), want.rgb)
    L = capi.lib()
    assert L.rt_mesh_replace(c._h, 7, tris.ctypes.data, 100, nodes.ctypes.data, int(nodes.shape[0])) == capi.RT_ERR_INVALID
    c.close()


@pytest.mark.parametrize("threshold", [1, 1 << 30])
def test_upload_soup_renders_the_golden_frames(capi, po, golden_hashes, threshold):
    """rt_mesh_upload_soup: file-order triangles in, BVH built and packed on the device (threshold 1) or on the
    host (threshold above every mesh): same triangle order back, same frames as the golden hashes."""
    import hashlib
    import torch
    c = capi.Context(0)
    c.set_option("device_build_min", threshold)
    meshes = po.default_meshes()
    for name, m in zip(NAMES, meshes):
        soup = po.load_obj(os.path.join(po.ASSETS, name + ".obj"))
        mid, post = c.upload_soup(soup, m.rgba, want_triangles=True)
        assert np.array_equal(post.view(np.uint32), np.ascontiguousarray(m.tris, np.float32).view(np.uint32))
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    pl = po.default_light()
    light = capi.Light(capi.Vec3(*pl.direction.tup()), capi.Vec3(*pl.color.tup()))
    w, h = 720, 540
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    hid = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
    for f in (0, 45, 95):
        tfs = [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in po.animation_transforms(f)]
        for variant in (0, 3):
            c.set_option("variant", variant)
            c.render(cam, light, tfs, w, h, rgb, hit_id=hid)
            c.sync()
            assert hashlib.sha256(rgb.cpu().numpy().tobytes()).hexdigest() == golden_hashes["frames"][f"{w}x{h}_f{f}"]["rgb"]
            assert hashlib.sha256(hid.cpu().numpy().tobytes()).hexdigest() == golden_hashes["frames"][f"{w}x{h}_f{f}"]["hit_id"]
    c.close()


def test_upload_soup_edge_cases_and_rebuild(capi, po):
    """Device path with a leaf too large for one reference (falls back to the host packer), a tiny soup, an empty
    one; rt_mesh_rebuild on the device path against the oracle."""
    import torch
    meshes = po.default_meshes()
    c = capi.Context(0)
    c.set_option("device_build_min", 1)
    one = soup_of(1, 7)
    big_leaf = np.concatenate([np.repeat(one, 300, axis=0), soup_of(500, 9)])
    drone_soup = po.load_obj(os.path.join(po.ASSETS, "drone.obj"))
    ids = [c.upload_soup(drone_soup, meshes[0].rgba), c.upload_soup(big_leaf, meshes[1].rgba), c.upload_soup(soup_of(3, 2), meshes[2].rgba),
           c.upload_soup(np.zeros((0, 24), np.float32), meshes[2].rgba)]
    assert ids == [0, 1, 2, 3]
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    pl = po.default_light()
    light = capi.Light(capi.Vec3(*pl.direction.tup()), capi.Vec3(*pl.color.tup()))
    w, h = 160, 90
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    hid = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
    soups = [drone_soup, big_leaf, soup_of(3, 2), np.zeros((0, 24), np.float32)]
    texs = [meshes[0].rgba, meshes[1].rgba, meshes[2].rgba, meshes[2].rgba]
    for frame in (0, 20):
        if frame:
            soups[0] = bake(drone_soup, 0.1 * frame, (1.0, 1.0, -0.5))
            post = c.rebuild_mesh(0, soups[0], want_triangles=True)
            assert np.array_equal(post.view(np.uint32), po.build_bvh(soups[0])[0].view(np.uint32))
        ref = [po.HostMesh(*po.build_bvh(s), t) for s, t in zip(soups, texs)]
        ptfs = [po.Transform(po.vec3(0, 0.2, 0), po.vec3(0, 0.3, 0)), po.Transform(po.vec3(-4, 0, 2), po.vec3(0, 0, 0)),
                po.Transform(po.vec3(-3, 0, -2), po.vec3(0.2, 0, 0)), po.Transform()]
        tfs = [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in ptfs]
        c.render(cam, light, tfs, w, h, rgb, hit_id=hid)
        c.sync()
        want = po.render_frame(ref, ptfs, po.default_camera(), po.default_light(), w, h)
        assert np.array_equal(hid.cpu().numpy(), want.hit_id) and np.array_equal(rgb.cpu().numpy(), want.rgb)
        assert len(set(np.unique(want.hit_id[want.hit_id >= 0] >> 24).tolist())) >= 2
    c.close()


def test_baked_grid_of_meshes_as_one_mesh_against_the_oracle(capi, po):
    """Copies of the treasure baked into one soup of half a million triangles, built and packed on the device,
    rendered by the plain frame loop: the frame of the reference loop on the oracle's own build of that soup."""
    import torch
    meshes = po.default_meshes()
    treasure = po.load_obj(os.path.join(po.ASSETS, "treasure.obj"))
    rng = np.random.default_rng(3)
    side, parts = 13, []
    for gx in range(side):
        for gz in range(side):
            parts.append(bake(treasure, float(rng.uniform(0, 6.28)), ((gx - 6) * 1.05, 0.45 + 0.1 * float(rng.uniform()), (gz - 6) * 1.05)))
    soup = np.concatenate(parts)
    assert soup.shape[0] > 500_000
    c = capi.Context(0)
    c.upload_mesh(meshes[0].tris, meshes[0].nodes, meshes[0].rgba)
    mid, post = c.upload_soup(soup, meshes[1].rgba, want_triangles=True)
    assert c.build_stats()["levels"] > 15                                      # went through the device build
    c.upload_mesh(meshes[2].tris, meshes[2].nodes, meshes[2].rgba)
    want_t, want_n = po.build_bvh(soup)
    assert np.array_equal(post.view(np.uint32), want_t.view(np.uint32))
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    pl = po.default_light()
    light = capi.Light(capi.Vec3(*pl.direction.tup()), capi.Vec3(*pl.color.tup()))
    w, h = 192, 108
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    hid = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
    t0 = po.animation_transforms(45)[0]
    c.render(cam, light, [capi.Transform(capi.Vec3(*t0.position.tup()), capi.Vec3(*t0.rotation.tup())), capi.Transform(), capi.Transform()],
             w, h, rgb, hit_id=hid)
    c.sync()
    want = po.render_frame([meshes[0], po.HostMesh(want_t, want_n, meshes[1].rgba), meshes[2]], [t0, po.Transform(), po.Transform()],
                           po.default_camera(), po.default_light(), w, h)
    assert np.array_equal(hid.cpu().numpy(), want.hit_id) and np.array_equal(rgb.cpu().numpy(), want.rgb)
    assert ((want.hit_id >> 24) == 1).mean() > 0.3
    c.close()
