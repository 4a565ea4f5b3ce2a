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
