"""Edge cases of the plain frame paths on the GPU, each against the CPU oracle's reference loop:
an empty scene, empty and single-leaf meshes, a leaf too large for one leaf reference (the reference
keeps a node as a leaf when its partition fails, bvh.c:51-55), coincident meshes (the earlier list
entry wins, scene.c:76), and the same through rt_render_frames and rt_render_frame_sampled."""
import numpy as np
import pytest

from test_host_emul import checker, tri_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from raytracer_amd import capi as m
    m.lib()
    return m


def camera_light(capi, po):
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    l = po.default_light()
    return cam, capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))


def conv(capi, tfs):
    return [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs]


def render_all_ways(capi, po, parts, tfs, w, h):
    """parts: [(triangles post-build, nodes, rgba)].  Renders with rt_render_frame, rt_render_frames (two frames)
    and rt_render_frame_sampled (1 sample, 1 light) and checks every result against the oracle."""
    import torch
    c = capi.Context(0)
    for t, n, r in parts:
        c.upload_mesh(t, n, r)
    cam, light = camera_light(capi, po)
    want = po.render_frame([po.HostMesh(np.ascontiguousarray(t, np.float32), n, r) for t, n, r in parts], tfs,
                           po.default_camera(), po.default_light(), w, h)
    ctf = conv(capi, tfs)
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    hid = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    two = torch.zeros((2, h, w, 3), dtype=torch.uint8, device="cuda:0")
    smp = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    c.render(cam, light, ctf, w, h, rgb, hit_id=hid)
    c.render_frames(cam, light, [ctf, ctf], w, h, two)
    c.render_sampled(cam, [light], ctf, w, h, 1, 0, smp)
    c.sync()
    assert np.array_equal(hid.cpu().numpy(), want.hit_id)
    assert np.array_equal(rgb.cpu().numpy(), want.rgb)
    assert np.array_equal(two[0].cpu().numpy(), want.rgb) and np.array_equal(two[1].cpu().numpy(), want.rgb)
    assert np.array_equal(smp.cpu().numpy(), want.rgb)
    c.close()
    return want


def test_empty_scene_is_background(capi, po):
    want = render_all_ways(capi, po, [], [], 70, 41)
    assert (want.rgb == 50).all() and (want.hit_id == -1).all()               # scene.c:147-150


def test_empty_and_single_leaf_meshes(capi, po, meshes):
    small = tri_rows([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[-1, -1, 0.5], [1, -1, 0.5], [0, 1, 0.5]]])
    post, nodes = po.build_bvh(small)
    e_post, e_nodes = po.build_bvh(np.zeros((0, 24), np.float32))
    parts = [(post, nodes, checker()), (e_post, e_nodes, checker()), (meshes[2].tris, meshes[2].nodes, meshes[2].rgba)]
    tfs = [po.Transform(po.vec3(0, 1, 0), po.vec3(0, 0.3, 0)), po.Transform(), po.Transform()]
    want = render_all_ways(capi, po, parts, tfs, 128, 72)
    ids = want.hit_id[want.hit_id >= 0] >> 24
    assert (ids == 0).any() and (ids == 2).any() and not (ids == 1).any()


def test_oversize_leaf_and_coincident_meshes(capi, po, meshes):
    """300 identical triangles stay one leaf in the reference; the first of them is the one reported.  The same
    mesh uploaded twice at the same place: every hit belongs to the first."""
    one = tri_rows([[[-1.5, -1, 0], [1.5, -1, 0], [0, 1.5, 0]]])
    post, nodes = po.build_bvh(np.repeat(one, 300, axis=0))
    assert nodes.shape[0] == 1 and nodes["count"][0] == 300
    parts = [(post, nodes, checker()), (post, nodes, checker()), (meshes[2].tris, meshes[2].nodes, meshes[2].rgba)]
    tf = po.Transform(po.vec3(0, 1.2, 0), po.vec3(0.1, 0.7, 0))
    want = render_all_ways(capi, po, parts, [tf, tf, po.Transform()], 160, 90)
    hit = want.hit_id[want.hit_id >= 0]
    assert ((hit >> 24) == 0).any() and not ((hit >> 24) == 1).any()
    assert set(np.unique(hit[(hit >> 24) == 0] & 0xFFFFFF).tolist()) == {0}


@pytest.mark.parametrize("size", [(240, 135), (64, 8), (50, 13), (1, 1)])
def test_strip_shares_equal_the_whole_frame(capi, po, meshes, size):
    """A frame rendered as N interleaved shares of eight-row strips (what N GPUs would do) and put back
    together equals the frame rendered in one go, for the plain and the sampled kernel, planes included."""
    import torch
    from raytracer_amd import sharding
    w, h = size
    c = capi.Context(0)
    for m in meshes:
        c.upload_mesh(m.tris, m.nodes, m.rgba)
    cam, light = camera_light(capi, po)
    l2 = capi.Light(capi.Vec3(-0.4, 0.8, 0.4), capi.Vec3(0.3, 0.5, 0.9))
    tfs = conv(capi, po.animation_transforms(45))
    dev = "cuda:0"
    full = {k: torch.zeros((h, w) + s, dtype=d, device=dev) for k, s, d in
            (("rgb", (3,), torch.uint8), ("hid", (), torch.int32), ("t", (), torch.float32), ("sh", (), torch.uint8),
             ("srgb", (3,), torch.uint8), ("shid", (), torch.int32))}
    torch.cuda.synchronize()
    c.render(cam, light, tfs, w, h, full["rgb"], hit_id=full["hid"], hit_t=full["t"], shadow=full["sh"])
    c.render_sampled(cam, [light, l2], tfs, w, h, 4, 7, full["srgb"], hit_id=full["shid"])
    c.sync()
    for world in (1, 2, 3, 5):
        parts = {k: [] for k in full}
        for r in range(world):
            rows = capi.strip_rows(h, world, r)
            assert rows == sharding.strip_rows(h, world, r)
            bufs = {k: torch.full((max(rows, 1),) + tuple(v.shape[1:]), 7, dtype=v.dtype, device=dev) for k, v in full.items()}
            torch.cuda.synchronize()
            c.render_strips(cam, light, tfs, w, h, world, r, bufs["rgb"], hit_id=bufs["hid"], hit_t=bufs["t"], shadow=bufs["sh"])
            c.render_sampled_strips(cam, [light, l2], tfs, w, h, 4, 7, world, r, bufs["srgb"], hit_id=bufs["shid"])
            c.sync()
            for k in full:
                parts[k].append(bufs[k])
        for k in full:
            got = sharding.assemble_strips(parts[k], h)
            assert torch.equal(got.view(torch.uint8), full[k].view(torch.uint8)), (k, world)
    # argument errors
    with pytest.raises(capi.RtError):
        c.render_strips(cam, light, tfs, w, h, 2, 2, full["rgb"])
    with pytest.raises(capi.RtError):
        c.render_strips(cam, light, tfs, w, h, 0, 0, full["rgb"])
    with pytest.raises(capi.RtError):
        c.render_sampled_strips(cam, [light], tfs, w, h, 2, 0, 3, -1, full["rgb"])
    c.close()


def test_limits_eight_meshes_and_sixteen_frames(capi, po, meshes):
    """The plain paths at their limits: RT_MAX_MESHES meshes in one scene, RT_MAX_BATCH frames in one launch;
    one mesh more is refused with a pointer to the instanced path."""
    import torch
    c = capi.Context(0)
    parts, tfs = [], []
    rng = np.random.default_rng(4)
    for k in range(capi.RT_MAX_MESHES):
        m = meshes[k % 3]
        parts.append(m)
        c.upload_mesh(m.tris, m.nodes, m.rgba)
        tfs.append(po.Transform(po.vec3(*(rng.uniform(-2, 2, 3) * [1, 0.3, 1] + [0, 0.6, 0])), po.vec3(0, float(rng.uniform(0, 6)), 0)) if k % 3 != 2
                   else po.Transform(po.vec3(0, -0.2 * (k // 3), 0), po.vec3(0, 0, 0)))
    cam, light = camera_light(capi, po)
    w, h = 200, 112
    n = capi.RT_MAX_BATCH
    frames = []
    for f in range(n):
        t = [po.Transform(po.vec3(x.position.x + 0.05 * f, x.position.y, x.position.z), x.rotation) for x in tfs]
        frames.append(t)
    buf = torch.zeros((n, h, w, 3), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    c.render_frames(cam, light, [conv(capi, t) for t in frames], w, h, buf)
    c.sync()
    got = buf.cpu().numpy()
    for f in (0, 7, n - 1):
        want = po.render_frame(parts, frames[f], po.default_camera(), po.default_light(), w, h)
        assert np.array_equal(got[f], want.rgb), f
        assert len(set(np.unique(want.hit_id[want.hit_id >= 0] >> 24).tolist())) >= 5
    c.upload_mesh(meshes[0].tris, meshes[0].nodes, meshes[0].rgba)            # a ninth mesh
    one = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(capi.RtError) as e:
        c.render(cam, light, conv(capi, frames[0] + [po.Transform()]), w, h, one)
    assert "instanced" in str(e.value)
    c.close()


def test_frames_in_flight_through_the_c_abi(capi, po, meshes):
    """rt_render_frame_host with option frames_in_flight: calls return before their frame has arrived, rt_sync(ctx, NULL)
    waits for all of them; pageable destinations, more frames than slots, a change of size in between, a call that
    asks for debug planes (always waits), and values outside 1..8 refused."""
    c = capi.Context(0)
    for m in meshes:
        c.upload_mesh(m.tris, m.nodes, m.rgba)
    cam, light = camera_light(capi, po)
    rng = np.random.default_rng(11)
    placements = [[po.Transform(po.vec3(*(rng.uniform(-1.5, 1.5, 3) * [1, 0.2, 1] + [0, 1, 0])), po.vec3(0, float(rng.uniform(0, 6)), 0)),
                   po.Transform(po.vec3(float(rng.uniform(-1, 1)), 0.5, 0), po.vec3(0, float(rng.uniform(0, 6)), 0)), po.Transform()]
                  for _ in range(7)]
    sizes = [(161, 90)] * 4 + [(96, 131)] * 3
    want = [c.render_host(cam, light, conv(capi, t), w, h, debug=False)[0] for t, (w, h) in zip(placements, sizes)]
    oracle = po.render_frame(list(meshes), placements[2], po.default_camera(), po.default_light(), *sizes[2])
    assert np.array_equal(want[2], oracle.rgb)
    assert c.get_option("frames_in_flight") == 1
    for bad in (0, 9, -1):
        with pytest.raises(capi.RtError):
            c.set_option("frames_in_flight", bad)
    for depth in (2, 3):
        c.set_option("frames_in_flight", depth)
        assert c.get_option("frames_in_flight") == depth
        got = [c.render_host(cam, light, conv(capi, t), w, h, debug=False)[0] for t, (w, h) in zip(placements, sizes)]
        rgb, hid, _, _ = c.render_host(cam, light, conv(capi, placements[2]), *sizes[2], debug=True)      # waits for its own frame
        assert np.array_equal(rgb, oracle.rgb) and np.array_equal(hid, oracle.hit_id)
        c.sync()
        for k in range(len(got)):
            assert np.array_equal(got[k], want[k]), (depth, k)
    c.set_option("frames_in_flight", 1)
    assert np.array_equal(c.render_host(cam, light, conv(capi, placements[0]), *sizes[0], debug=False)[0], want[0])
    c.close()


def test_c_abi_walk_through_from_a_compiled_program():
    """tests/host_emul/abi_check.cpp (built by `make`): a C++ program on include/rt_mi355.h and the HIP runtime only --
    uploads by every route, all kernel variants, batches, strips, sampled and instanced frames with changing lists,
    rebuilt meshes, frames in flight into pinned and pageable memory, the upscale, refused arguments -- checks every
    status and that the different routes to one frame give the same bytes."""
    import json
    import os
    import subprocess
    from conftest import REPO
    exe = os.path.join(REPO, "tests", "host_emul", "abi_check")
    assert os.path.exists(exe), "run `make`"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"errors": 0}


# ------------------------------------------------------------------ tolerant OBJ reader, end to end on the GPU
def _terrain_obj(n=12):
    """A height field written the ways the reference's reader cannot take (mesh.c:41-76 reads triangles with
    v/vt/vn corners only): quads and five-corner faces, corners without texture coordinates (v//vn), and faces
    that count from the end of the lists.  Every number is a multiple of 1/64, so any parser reads the same
    float.  Returns (text, the triangles a fan over each face gives, in file order, [k, 24] float32)."""
    rng = np.random.default_rng(11)
    q = lambda a: np.round(np.asarray(a, np.float64) * 64) / 64          # noqa: E731
    xs = q(np.linspace(-2.0, 2.0, n + 1))
    hgt = q(0.3 + 0.25 * rng.random((n + 1, n + 1)))
    v = np.array([[xs[i], hgt[j, i], xs[j]] for j in range(n + 1) for i in range(n + 1)])
    vt = q(np.array([[i / n, j / n] for j in range(n + 1) for i in range(n + 1)]))
    vn = q(np.array([[0.0, 1.0, 0.0], [0.25, 0.9375, 0.0], [0.0, 0.9375, -0.25], [-0.125, 0.96875, 0.125]]))
    lines = ["# terrain"] + ["v %r %r %r" % tuple(map(float, p)) for p in v]
    lines += ["vt %r %r" % tuple(map(float, p)) for p in vt] + ["vn %r %r %r" % tuple(map(float, p)) for p in vn]
    nv, nt, nn = len(v), len(vt), len(vn)
    tris = []

    def corner(iv, it, inn):
        return v[iv], (vt[it] if it is not None else np.zeros(2)), vn[inn]

    def emit(corners, style):
        toks = []
        for iv, it, inn in corners:
            if style == "neg":
                toks.append("%d/%d/%d" % (iv - nv, it - nt, inn - nn))
            elif style == "novt":
                toks.append("%d//%d" % (iv + 1, inn + 1))
            else:
                toks.append("%d/%d/%d" % (iv + 1, it + 1, inn + 1))
        lines.append("f " + " ".join(toks))
        cs = [corner(iv, None if style == "novt" else it, inn) for iv, it, inn in corners]
        for k in range(1, len(cs) - 1):                                     # the fan: corners 0, k, k+1
            a, b, c = cs[0], cs[k], cs[k + 1]
            tris.append(np.concatenate([a[0], b[0], c[0], a[1], b[1], c[1], a[2], b[2], c[2]]))

    idx = lambda i, j: j * (n + 1) + i                                       # noqa: E731
    for j in range(n):
        for i in range(n):
            k = (i + j) % 4
            quad = [(idx(i, j), idx(i, j), k), (idx(i, j + 1), idx(i, j + 1), k), (idx(i + 1, j + 1), idx(i + 1, j + 1), k),
                    (idx(i + 1, j), idx(i + 1, j), k)]
            style = ("plain", "neg", "novt")[(i * 7 + j * 3) % 3]
            if i + 2 <= n and (i + j) % 5 == 0 and i % 2 == 0:               # a five-corner face over this cell and half of the next
                emit(quad[:3] + [(idx(i + 2, j + 1), idx(i + 2, j + 1), k), quad[3]], style)
            else:
                emit(quad, style)
    return "\n".join(lines) + "\n", np.array(tris, np.float32)


def test_tolerant_obj_through_create_mesh_renders_like_the_same_triangles_uploaded(capi, po, tmp_path):
    """create_mesh in tolerant mode on quads, five-corner faces, v//vn corners and negative indices -> render_scene on
    the GPU, against (a) the same triangles, fanned out here, uploaded as a soup through the C ABI and (b) the oracle's
    reference loop on its own build of that soup.  The reference's reader has no output for such a file (its sscanf
    pattern skips these faces), so the reader's semantics are pinned by this fan-out, not by the reference."""
    import ctypes as C
    import os
    from raytracer_amd import host_api as H
    text, soup = _terrain_obj()
    obj = tmp_path / "terrain.obj"
    obj.write_text(text)
    L = H.lib()
    w, h = 320, 180
    L.rt_host_set_obj_mode(1)
    try:
        m = L.create_mesh(str(obj).encode(), os.path.join(H.ASSETS, "ground.webp").encode())
    finally:
        L.rt_host_set_obj_mode(0)
    assert m.bvh.root and int(m.triangle_count) == soup.shape[0]
    parsed = H.mesh_triangles(m)
    key = lambda a: sorted(r.tobytes() for r in a)                           # noqa: E731
    assert key(parsed) == key(soup), "the reader's triangles are not the fan over each face"
    rgba = H.mesh_texture(m)
    scene = L.create_scene(w, h, 1000, 1, 1.0)
    L.set_scene_camera(C.byref(scene), H.Vec3(-3, 3, -3), H.Vec3(0, 0, 0), H.Vec3(0, 1, 0), 60.0)
    L.set_scene_light(C.byref(scene), H.Vec3(1, 1, -1), H.Vec3(1.4, 1.4, 1.4))
    L.add_mesh_to_scene(C.byref(scene), m)
    L.set_mesh_rotation(C.byref(scene.meshes[0]), H.Vec3(0.0, 0.5, 0.0))
    L.render_scene(C.byref(scene))
    assert L.rt_host_last_status(C.byref(scene)) == 0
    got_rgb = H.frame_array(scene, 0)
    hid, ht, sh = np.zeros((h, w), np.int32), np.zeros((h, w), np.float32), np.zeros((h, w), np.uint8)
    assert L.rt_host_render_debug(C.byref(scene), hid.ctypes.data, ht.ctypes.data, sh.ctypes.data) == 0
    L.destroy_mesh(C.byref(m))
    L.destroy_scene(C.byref(scene))

    tf = capi.Transform(capi.Vec3(0, 0, 0), capi.Vec3(0.0, 0.5, 0.0))
    cam, light = camera_light(capi, po)
    with capi.Context(0) as c:
        c.upload_soup(soup, rgba)
        rgb2, hid2, ht2, sh2 = c.render_host(cam, light, [tf], w, h)
    assert np.array_equal(got_rgb, rgb2) and np.array_equal(hid, hid2) and np.array_equal(sh, sh2)
    assert np.array_equal(ht.view(np.uint32), ht2.view(np.uint32))
    t_post, nodes = po.build_bvh(soup)
    want = po.render_frame([po.HostMesh(t_post, nodes, rgba)], [po.Transform(po.vec3(0, 0, 0), po.vec3(0.0, 0.5, 0.0))],
                           po.default_camera(), po.default_light(), w, h)
    assert np.array_equal(got_rgb, want.rgb) and np.array_equal(hid, want.hit_id) and np.array_equal(sh, want.shadow)
    assert np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32))
    assert int((hid >= 0).sum()) > w * h // 10 and len(np.unique(hid)) > 100


def test_render_scene_sees_a_bvh_rebuilt_in_place(capi, po):
    """The reference's render_scene reads the triangles as they are at the call (scene.c:72-81).  A caller that deforms
    mesh.triangles in place and rebuilds with destroy_bvh + create_bvh usually gets the old root address back from
    malloc; the host library must still upload the new geometry.  Also: triangles edited in place WITHOUT a rebuild
    are picked up after rt_host_invalidate_scene."""
    import ctypes as C
    import os
    from raytracer_amd import host_api as H
    L = H.lib()
    w, h = 200, 120
    m = L.create_mesh(os.path.join(H.ASSETS, "drone.obj").encode(), os.path.join(H.ASSETS, "drone.webp").encode())
    rgba = H.mesh_texture(m)
    scene = L.create_scene(w, h, 1000, 1, 1.0)
    L.set_scene_camera(C.byref(scene), H.Vec3(-1.5, 1.5, -1.5), H.Vec3(0, 0.3, 0), H.Vec3(0, 1, 0), 60.0)
    L.set_scene_light(C.byref(scene), H.Vec3(1, 1, -1), H.Vec3(1.4, 1.4, 1.4))
    L.add_mesh_to_scene(C.byref(scene), m)
    ident = [po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]

    def oracle_now():
        soup = H.mesh_triangles(scene.meshes[0])
        nodes = np.zeros(2 * soup.shape[0] + 1, capi.NODE_DTYPE)
        used = C.c_int32(0)
        assert capi.lib().rt_bvh_flatten_tree(C.cast(scene.meshes[0].bvh.root, C.c_void_p), nodes.ctypes.data, nodes.shape[0], C.byref(used)) == 0
        cam = po.Camera(po.vec3(-1.5, 1.5, -1.5), po.vec3(0, 0.3, 0), po.vec3(0, 1, 0), 60.0)
        return po.render_frame([po.HostMesh(soup, nodes[:used.value].copy(), rgba)], ident, cam, po.default_light(), w, h).rgb

    L.render_scene(C.byref(scene))
    first = H.frame_array(scene, 0)
    assert np.array_equal(first, oracle_now())
    # deform in place (stretch in y), rebuild the tree over the same array
    n = int(scene.meshes[0].triangle_count)
    tri = np.ctypeslib.as_array(C.cast(scene.meshes[0].triangles, C.POINTER(C.c_float)), shape=(n, 24))
    old_root = C.cast(scene.meshes[0].bvh.root, C.c_void_p).value
    tri[:, [1, 4, 7]] *= np.float32(1.5)
    L.destroy_bvh(C.byref(scene.meshes[0].bvh))
    scene.meshes[0].bvh = L.create_bvh(scene.meshes[0].triangles, n)
    same_address = C.cast(scene.meshes[0].bvh.root, C.c_void_p).value == old_root
    L.render_scene(C.byref(scene))
    second = H.frame_array(scene, 0)
    assert not np.array_equal(second, first)
    assert np.array_equal(second, oracle_now()), f"stale geometry rendered (root address reused: {same_address})"
    # in place without a rebuild: texture coordinates shifted (the tree stays valid for the positions)
    tri[:, 9:15] += np.float32(0.25)
    L.rt_host_invalidate_scene(C.byref(scene))
    L.render_scene(C.byref(scene))
    third = H.frame_array(scene, 0)
    assert np.array_equal(third, oracle_now()) and not np.array_equal(third, second)
    m.bvh = scene.meshes[0].bvh                    # destroy_mesh frees the tree that is alive now
    L.destroy_mesh(C.byref(m))
    L.destroy_scene(C.byref(scene))


def test_culling_keeps_the_reference_result_on_adversarial_scenes_gpu(capi, po):
    """The scenes of tests/adversarial.py on the GPU: default mode (ordered, distance-culled) and
    RT_FLAG_REFERENCE_ORDER against the oracle's reference loop, every plane bit for bit; and through
    rt_render_frames (the throughput kernel) for the RGB plane."""
    import torch
    import adversarial
    tex = checker()
    cam0, light = camera_light(capi, po)
    n = 0
    for name, soups, tfs, cam, w, h in adversarial.scenes(po, seed=5):
        meshes = adversarial.build(po, soups, tex)
        want = po.render_frame(meshes, tfs, cam, po.default_light(), w, h)
        ccam = capi.Camera(capi.Vec3(*cam.position.tup()), capi.Vec3(*cam.look_at.tup()), capi.Vec3(*cam.up.tup()), cam.fov)
        with capi.Context(0) as c:
            for m in meshes:
                c.upload_mesh(m.tris, m.nodes, m.rgba)
            ctf = conv(capi, tfs)
            for flags in (0, capi.RT_FLAG_REFERENCE_ORDER):
                rgb, hid, ht, sh = c.render_host(ccam, light, ctf, w, h, flags=flags)
                mode = "reference order" if flags else "default"
                assert np.array_equal(hid, want.hit_id), f"{name}: {(hid != want.hit_id).sum()} hit ids differ ({mode})"
                assert np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32)), f"{name}: t ({mode})"
                assert np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb), f"{name}: shading ({mode})"
            out = torch.zeros((2, h, w, 3), dtype=torch.uint8, device="cuda:0")
            c.render_frames(ccam, light, [ctf, ctf], w, h, out)
            c.sync()
            assert np.array_equal(out[0].cpu().numpy(), want.rgb) and np.array_equal(out[1].cpu().numpy(), want.rgb), name
        n += 1
    assert n >= 20


def test_degenerate_cameras_lights_and_placements_on_the_gpu(capi, po, meshes):
    """tests/adversarial.py::degenerate_views (zero-length and NaN camera bases, fields of view of 0 / 180 / 270 degrees, zero, NaN
    and axis-parallel lights, placements with NaN angles or at infinity): rt_render_frame in both traversal modes, rt_render_frames,
    rt_render_frame_sampled (one sample, one light) and rt_render_frame_instanced against the oracle's reference loop, every
    plane bit for bit (NaN distances included).  The reference reports no error on any of them (scene.c:52-160 has no checks)."""
    import torch
    import adversarial
    w, h = 96, 54
    n = 0
    with capi.Context(0) as c:
        for m in meshes:
            c.upload_mesh(m.tris, m.nodes, m.rgba)
        for name, tfs, cam, l in adversarial.degenerate_views(po):
            want = po.render_frame(meshes, tfs, cam, l, w, h)
            ccam = capi.Camera(capi.Vec3(*cam.position.tup()), capi.Vec3(*cam.look_at.tup()), capi.Vec3(*cam.up.tup()), cam.fov)
            light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
            ctf = conv(capi, tfs)
            for flags in (0, capi.RT_FLAG_REFERENCE_ORDER):
                rgb, hid, ht, sh = c.render_host(ccam, light, ctf, w, h, flags=flags)
                mode = "reference order" if flags else "default"
                assert np.array_equal(hid, want.hit_id), f"{name}: {(hid != want.hit_id).sum()} hit ids differ ({mode})"
                assert np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32)), f"{name}: t ({mode})"
                assert np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb), f"{name}: shading ({mode})"
            for variant in range(c.get_option("n_variants")):   # persistent, cooperative, two-pass and lean kernels as well
                c.set_option("variant", variant)
                rgb, hid, ht, sh = c.render_host(ccam, light, ctf, w, h)
                assert np.array_equal(hid, want.hit_id) and np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32)), f"{name}: variant {variant}"
                assert np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb), f"{name}: variant {variant}"
                # without the planes a call sends no shadow ray where it cannot change the colour: same bytes
                rgb_only, _, _, _ = c.render_host(ccam, light, ctf, w, h, debug=False)
                assert np.array_equal(rgb_only, want.rgb), f"{name}: variant {variant}, colour only"
            c.set_option("variant", c.get_option("default_variant"))
            out = torch.zeros((2, h, w, 3), dtype=torch.uint8, device="cuda:0")
            smp = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
            ins = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
            hi = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
            c.render_frames(ccam, light, [ctf, ctf], w, h, out)
            c.render_sampled(ccam, [light], ctf, w, h, 1, 0, smp)
            c.render_instanced(ccam, light, capi.Context.make_instances([0, 1, 2], ctf), w, h, ins, hit_instance=hi)
            c.sync()
            assert np.array_equal(out[0].cpu().numpy(), want.rgb) and np.array_equal(out[1].cpu().numpy(), want.rgb), f"{name}: batched frames"
            assert np.array_equal(smp.cpu().numpy(), want.rgb), f"{name}: sampled frame"
            assert np.array_equal(ins.cpu().numpy(), want.rgb), f"{name}: instanced frame"
            assert np.array_equal(hi.cpu().numpy(), np.where(want.hit_id >= 0, want.hit_id >> 24, -1)), f"{name}: instance ids"
            n += 1
    assert n >= 20


def test_meshes_with_nan_and_infinite_vertices_render_like_the_reference(capi, po):
    """tests/adversarial.py::nonfinite_soups as meshes (boxes with NaN and infinite bounds, triangles the Moeller-Trumbore test can
    only reject): both traversal modes against the oracle's reference loop, every plane."""
    import adversarial
    tex = checker()
    cam, light = camera_light(capi, po)
    ident = [po.Transform(po.vec3(0, 0, 0), po.vec3(0.2, 0.4, 0.1))]
    n = hits = 0
    for name, soup in adversarial.nonfinite_soups(sizes=(7, 60, 500)):
        t, nodes = capi.build_bvh(soup)
        want = po.render_frame([po.HostMesh(t, nodes, tex)], ident, po.default_camera(), po.default_light(), 96, 54)
        with capi.Context(0) as c:
            c.upload_mesh(t, nodes, tex)
            for flags in (0, capi.RT_FLAG_REFERENCE_ORDER):
                rgb, hid, ht, sh = c.render_host(cam, light, conv(capi, ident), 96, 54, flags=flags)
                assert np.array_equal(hid, want.hit_id), f"{name}: hit ids (flags {flags})"
                assert np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32)), f"{name}: t (flags {flags})"
                assert np.array_equal(sh, want.shadow) and np.array_equal(rgb, want.rgb), f"{name}: shading (flags {flags})"
        hits += int((want.hit_id >= 0).sum())
        n += 1
    assert n >= 30 and hits > 5000


def test_odd_texture_shapes_and_extreme_frame_shapes(capi, po, meshes):
    """Textures of 1 x 1, one row, one column and very oblong shapes with texture coordinates outside [0, 1) (mesh.c:138-149
    multiplies by size - 1, which is 0 for a size of 1); frames one pixel wide or high, 16384 x 1, 3 x 2051, sizes one above and
    below a tile -- rt_render_frame (all planes), rt_render_frames, rt_render_frame_sampled and rt_render_frame_instanced."""
    import torch
    import adversarial
    rng = np.random.default_rng(2)
    cam, light = camera_light(capi, po)
    rows = np.asarray(adversarial._blob(rng, np.zeros(3), 0.9, 300, size=0.6), np.float32)
    rows[:, 9:15] = rng.uniform(-2, 3, size=(300, 6)).astype(np.float32)
    t, nodes = capi.build_bvh(rows)
    ident = [po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]
    for shape in ((1, 1), (1, 7), (5, 1), (2, 2), (3, 1000), (1000, 3)):
        tex = rng.integers(0, 256, size=shape + (4,), dtype=np.uint8)
        want = po.render_frame([po.HostMesh(t, nodes, tex)], ident, po.default_camera(), po.default_light(), 96, 54)
        with capi.Context(0) as c:
            c.upload_mesh(t, nodes, tex)
            rgb, hid, _, _ = c.render_host(cam, light, conv(capi, ident), 96, 54)
        assert np.array_equal(rgb, want.rgb) and np.array_equal(hid, want.hit_id), f"texture of shape {shape}"
    tfs = po.animation_transforms(33)
    ctf = conv(capi, tfs)
    with capi.Context(0) as c:
        for m in meshes:
            c.upload_mesh(m.tris, m.nodes, m.rgba)
        for w, h in ((1, 97), (97, 1), (8, 8), (9, 9), (7, 7), (4099, 3), (3, 2051), (65, 63), (16384, 1)):
            want = po.render_frame(meshes, tfs, po.default_camera(), po.default_light(), w, h)
            rgb, hid, ht, sh = c.render_host(cam, light, ctf, w, h)
            assert np.array_equal(rgb, want.rgb) and np.array_equal(hid, want.hit_id) and np.array_equal(sh, want.shadow), (w, h)
            assert np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32)), (w, h)
            three = torch.zeros((3, h, w, 3), dtype=torch.uint8, device="cuda:0")
            smp = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
            ins = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            c.render_frames(cam, light, [ctf, ctf, ctf], w, h, three)
            c.render_sampled(cam, [light], ctf, w, h, 1, 0, smp)
            c.render_instanced(cam, light, capi.Context.make_instances([0, 1, 2], ctf), w, h, ins)
            c.sync()
            assert all(np.array_equal(three[k].cpu().numpy(), want.rgb) for k in range(3)), (w, h)
            assert np.array_equal(smp.cpu().numpy(), want.rgb) and np.array_equal(ins.cpu().numpy(), want.rgb), (w, h)
