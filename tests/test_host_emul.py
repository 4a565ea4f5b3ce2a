"""The product's host logic (scene packer, per-frame constants) and its pixel pipeline, compiled for
the host (tests/host_emul), against the CPU oracle.  Covers what the GPU tests cover, on CPU, plus
the packer's edge cases: single-leaf and empty meshes, leaves larger than the reference's 4
triangles, ties, malformed trees.  CPU only."""
import numpy as np
import pytest

import emul
from conftest import bits


def node_array(po, rows):
    a = np.zeros(len(rows), dtype=po.NODE_DTYPE)
    for i, (lo, hi, left, right, start, count) in enumerate(rows):
        a[i] = (lo, hi, left, right, start, count)
    return a


def tri_rows(v, uv=None, n=None):
    """[k,3,3] vertex positions -> [k,24] reference triangles (v0 v1 v2 t0 t1 t2 n0 n1 n2)."""
    v = np.asarray(v, np.float32)
    k = v.shape[0]
    out = np.zeros((k, 24), np.float32)
    out[:, 0:9] = v.reshape(k, 9)
    out[:, 9:15] = (0.1, 0.1, 0.9, 0.1, 0.5, 0.9) if uv is None else uv
    nn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-20)
    out[:, 15:24] = np.tile(nn, 3) if n is None else n
    return out


def bounds(tris):
    p = tris[:, 0:9].reshape(-1, 3)
    return p.min(axis=0), p.max(axis=0)


def checker(n=8):
    t = np.zeros((n, n, 4), np.uint8)
    t[..., 3] = 255
    for y in range(n):
        for x in range(n):
            t[y, x, :3] = (40 + 25 * x, 30 + 20 * y, 200 - 10 * (x + y))
    return t


def compare(po, em_scene, meshes, tfs, cam, light, w, h, full=False):
    want = po.render_frame(meshes, tfs, cam, light, w, h)
    got = em_scene.render(cam, light, tfs, w, h, full=full)
    assert np.array_equal(got["hit_id"], want.hit_id), f"{(got['hit_id'] != want.hit_id).sum()} ids differ"
    assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t))
    assert np.array_equal(got["shadow"], want.shadow)
    assert np.array_equal(got["rgb"], want.rgb)
    return got, want


@pytest.fixture(scope="module", params=[False, True], ids=["reference-tree", "regrouped-tree"])
def em_default(meshes, request):
    """The reference scene, packed with the reference's tree only or with the regrouped tree in front of it."""
    s = emul.Scene(regroup=request.param)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    yield s
    s.close()


# ------------------------------------------------------------------------------ the reference scene

@pytest.mark.parametrize("frame", [0, 45, 95])
def test_pipeline_equals_oracle_on_golden_frames(po, meshes, em_default, golden_frames, frame):
    got, want = compare(po, em_default, meshes, po.animation_transforms(frame), po.default_camera(), po.default_light(), 240, 135)
    assert np.array_equal(got["rgb"], golden_frames[f"f{frame}_rgb"])
    assert np.array_equal(got["hit_id"], golden_frames[f"f{frame}_hit_id"])


def test_reference_order_mode_does_the_reference_work_exactly(po, meshes, em_default, request):
    tfs = po.animation_transforms(45)
    got, want = compare(po, em_default, meshes, tfs, po.default_camera(), po.default_light(), 200, 112, full=True)
    assert {k: got["counters"][k] for k in want.counters} == want.counters
    # culled mode: same image, never more work, per pixel
    fast = em_default.render(po.default_camera(), po.default_light(), tfs, 200, 112, full=False, per_pixel=True)
    slow = em_default.render(po.default_camera(), po.default_light(), tfs, 200, 112, full=True, per_pixel=True)
    assert np.array_equal(fast["rgb"], slow["rgb"]) and np.array_equal(fast["hit_id"], slow["hit_id"])
    if "regrouped" in request.node.name:
        # another tree over the same leaves: fewer box tests in total (not for every single pixel), the same leaves
        assert fast["counters"]["slab_tests"] < 0.9 * slow["counters"]["slab_tests"]
        assert fast["counters"]["tri_tests"] <= slow["counters"]["tri_tests"]
    else:
        assert (fast["px_slabs"] <= slow["px_slabs"]).all() and (fast["px_tris"] <= slow["px_tris"]).all()
        assert fast["counters"]["slab_tests"] < slow["counters"]["slab_tests"]


def test_stack_use_stays_within_tree_depth(po, meshes, em_default):
    deepest = max(em_default.info(i)["depth"] for i in range(3))
    for full in (False, True):
        r = em_default.render(po.default_camera(), po.default_light(), po.animation_transforms(10), 160, 90, full=full)
        assert 0 < r["stack_high"] <= deepest + 2


@pytest.mark.parametrize("size", [(1, 1), (3, 2), (17, 9), (65, 33)])
def test_ragged_sizes(po, meshes, em_default, size):
    compare(po, em_default, meshes, po.animation_transforms(77), po.default_camera(), po.default_light(), *size)


def test_random_placements_cameras_lights(po, meshes, em_default):
    rng = np.random.default_rng(5)
    for _ in range(8):
        tfs = [po.Transform(po.vec3(*rng.uniform(-1.5, 1.5, 3)), po.vec3(*rng.uniform(-3.2, 3.2, 3))) for _ in range(2)]
        tfs.append(po.Transform(po.vec3(0, 0, 0), po.vec3(*rng.uniform(-0.05, 0.05, 3))))
        eye = rng.uniform(-4, 4, 3)
        eye[1] = abs(eye[1]) + 1.0
        cam = po.Camera(po.vec3(*eye), po.vec3(0, 0.3, 0), po.vec3(0, 1, 0), float(rng.uniform(30, 90)))
        light = po.lib().orc_make_light(po.vec3(*rng.uniform(-1, 1, 3)), po.vec3(1.1, 0.9, 1.3))
        compare(po, em_default, meshes, tfs, cam, light, 96, 54)


def test_camera_inside_a_mesh_and_axis_aligned_rays(po, meshes, em_default):
    """Camera at the treasure's centre looking along an axis: rays start inside boxes, several direction
    components are exactly zero (1/0 = inf in the slab test)."""
    tfs = [po.Transform(po.vec3(0, 5, 0), po.vec3(0, 0, 0)), po.Transform(po.vec3(0, 0.5, 0), po.vec3(0, 0, 0)),
           po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]
    cam = po.Camera(po.vec3(0, 0.5, 0), po.vec3(0, 0.5, 1), po.vec3(0, 1, 0), 90.0)
    compare(po, em_default, meshes, tfs, cam, po.lib().orc_make_light(po.vec3(0, 1, 0), po.vec3(1, 1, 1)), 65, 65)
    cam = po.Camera(po.vec3(0, 4, 0), po.vec3(0, 0, 0), po.vec3(0, 0, 1), 40.0)           # straight down
    compare(po, em_default, meshes, tfs, cam, po.lib().orc_make_light(po.vec3(0, 1, 0), po.vec3(1, 1, 1)), 33, 33)


def test_per_frame_constants_equal_the_oracles(po):
    import ctypes as C
    rng = np.random.default_rng(9)
    cam, light = po.default_camera(), po.default_light()
    for k in range(64):
        pos = rng.uniform(-3, 3, 3) if k >= 8 else np.zeros(3)
        rot = rng.uniform(-4, 4, 3) if k >= 4 else np.zeros(3)
        tf = po.Transform(po.vec3(*pos), po.vec3(*rot))
        inv, nrm = po.mesh_matrices(tf)
        inv16, nrm9 = np.zeros(16, np.float32), np.zeros(9, np.float32)
        cam_o, lit_d, mode = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_uint32(9)
        tf2, cam2, light2 = emul.conv(tf, emul.Transform), emul.conv(cam, emul.Camera), emul.conv(light, emul.Light)
        emul.lib().emul_placement(C.byref(tf2), C.byref(cam2), C.byref(light2), inv16.ctypes.data, nrm9.ctypes.data,
                                  cam_o.ctypes.data, lit_d.ctypes.data, C.byref(mode))
        assert np.array_equal(bits(inv16.reshape(4, 4)), bits(inv))
        assert np.array_equal(bits(nrm9.reshape(3, 3)), bits(nrm[:3, :3]))
        # camera origin in mesh space = transform_ray's origin (ray.c:5-16)
        r = po.lib().orc_transform_ray(po.Ray(cam.position, po.vec3(0, 0, 1)), tf)
        assert np.array_equal(bits(cam_o), bits(np.array(r.origin.tup(), np.float32)))
        assert mode.value in (0, 1, 2)


# ------------------------------------------------------------------------------ packer edge cases

def scene_of(po, parts):
    """parts: list of (tris[k,24], nodes, rgba).  Returns (emul scene, oracle meshes)."""
    s = emul.Scene()
    ms = []
    for tris, nodes, rgba in parts:
        assert s.add(tris, nodes, rgba) == 0, s.error()
        ms.append(po.HostMesh(np.ascontiguousarray(tris, np.float32), nodes, rgba))
    return s, ms


def front_camera(po):
    return po.Camera(po.vec3(0, 0, -4), po.vec3(0, 0, 0), po.vec3(0, 1, 0), 50.0)


def identity(po):
    return po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))


def test_mesh_that_is_a_single_leaf(po, meshes):
    tris = tri_rows([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[-1, -1, 1], [1, -1, 1], [0, 1, 1.5]], [[0.2, 0, -1], [0.9, 0, -1], [0.5, 0.7, -1]]])
    post, nodes = po.build_bvh(tris)
    assert nodes.shape[0] == 1
    s, ms = scene_of(po, [(post, nodes, checker()), (meshes[2].tris, meshes[2].nodes, meshes[2].rgba)])
    assert s.info(0)["pairs"] == 0 and s.info(0)["root_ref"] & 0x80000000
    tfs = [po.Transform(po.vec3(0, 1, 0), po.vec3(0.3, 0.2, 0.1)), identity(po)]
    got, want = compare(po, s, ms, tfs, po.default_camera(), po.default_light(), 120, 68)
    assert (want.hit_id >> 24 == 0).any() and (want.shadow == 1).any()
    s.close()


def test_empty_mesh_is_accepted_and_never_hit(po, meshes):
    empty_tris = np.zeros((0, 24), np.float32)
    post, nodes = po.build_bvh(empty_tris)               # one node, count 0, inverted (empty) box like create_empty_aabb
    assert nodes.shape[0] == 1 and nodes["count"][0] == 0
    s, ms = scene_of(po, [(post, nodes, checker()), (meshes[0].tris, meshes[0].nodes, meshes[0].rgba)])
    tfs = [identity(po), po.animation_transforms(3)[0]]
    got, want = compare(po, s, ms, tfs, po.default_camera(), po.default_light(), 100, 60)
    assert not (want.hit_id >> 24 == 0)[want.hit_id >= 0].any()
    s.close()


@pytest.mark.parametrize("count", [5, 9, 127, 128, 300])
def test_leaf_larger_than_four_triangles(po, count):
    """The reference keeps a node as one leaf when its partition fails (bvh.c:51-55), so leaves of any
    size can occur; the packer splits them into chains of at most 127."""
    rng = np.random.default_rng(count)
    c = rng.uniform(-1, 1, size=(count, 1, 3)) * (1.2, 1.2, 0.6)
    v = c + rng.uniform(-0.25, 0.25, size=(count, 3, 3))
    tris = tri_rows(v, uv=rng.uniform(0, 1, size=(count, 6)).astype(np.float32))
    lo, hi = bounds(tris)
    nodes = node_array(po, [(lo, hi, -1, -1, 0, count)])
    s, ms = scene_of(po, [(tris, nodes, checker())])
    got, want = compare(po, s, ms, [po.Transform(po.vec3(0, 0, 0), po.vec3(0.2, 0.4, 0))], front_camera(po), po.default_light(), 96, 96)
    assert (want.hit_id >= 0).sum() > 50
    # the work is the reference's in reference-order mode: one triangle test per triangle of the leaf
    full = s.render(front_camera(po), po.default_light(), [po.Transform(po.vec3(0, 0, 0), po.vec3(0.2, 0.4, 0))], 96, 96, full=True)
    if count <= 127:
        assert {k: full["counters"][k] for k in want.counters} == want.counters
    else:       # the chain's own links are extra box tests; the triangle tests are still the reference's
        assert full["counters"]["tri_tests"] == want.counters["tri_tests"]
        assert full["counters"]["slab_tests"] >= want.counters["slab_tests"]
    s.close()


def test_oversize_leaf_inside_a_tree(po):
    rng = np.random.default_rng(21)
    left = tri_rows(rng.uniform(-0.3, 0.3, size=(3, 3, 3)) + (-0.8, 0, 0))
    right = tri_rows(rng.uniform(-0.4, 0.4, size=(140, 3, 3)) + (0.7, 0, 0.2))
    tris = np.concatenate([left, right])
    (llo, lhi), (rlo, rhi) = bounds(left), bounds(right)
    nodes = node_array(po, [(np.minimum(llo, rlo), np.maximum(lhi, rhi), 1, 2, 0, 143), (llo, lhi, -1, -1, 0, 3), (rlo, rhi, -1, -1, 3, 140)])
    s, ms = scene_of(po, [(tris, nodes, checker())])
    compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 128, 72)
    compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 128, 72, full=True)
    s.close()


def test_ties_within_a_leaf_between_siblings_and_between_meshes(po):
    """Coincident triangles hit at exactly the same t.  The reference keeps the first one tested inside a
    leaf (strict <, bvh.c:83-92), lets the right sibling replace the left one's hit (bvh.c:99-107 test with <=
    through the closest_t update order), and the earlier mesh wins between meshes (scene.c:72-88).  Whatever
    the rule, the product must reproduce the oracle's choice."""
    a = [[-1, -1, 0], [1, -1, 0], [0, 1, 0]]
    b = [[-0.5, -1, 0.5], [1.5, -1, 0.5], [0.5, 1, 0.5]]
    # same leaf: indices 0 and 1 coincide
    tris = tri_rows([a, a, b, b], uv=np.array([[0.1, 0.1, 0.2, 0.1, 0.1, 0.2], [0.8, 0.8, 0.9, 0.8, 0.8, 0.9],
                                             [0.1, 0.8, 0.2, 0.8, 0.1, 0.9], [0.8, 0.1, 0.9, 0.1, 0.8, 0.2]], np.float32))
    lo, hi = bounds(tris)
    one_leaf = node_array(po, [(lo, hi, -1, -1, 0, 4)])
    s, ms = scene_of(po, [(tris, one_leaf, checker())])
    got, want = compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 64, 64)
    assert set(np.unique(want.hit_id[want.hit_id >= 0]).tolist()) <= {0, 2}      # first of each coincident pair
    s.close()
    # siblings: triangle 0 in the left leaf, its twin (index 1) in the right leaf
    tris2 = tri_rows([a, a], uv=np.array([[0.1, 0.1, 0.2, 0.1, 0.1, 0.2], [0.8, 0.8, 0.9, 0.8, 0.8, 0.9]], np.float32))
    lo, hi = bounds(tris2)
    sib = node_array(po, [(lo, hi, 1, 2, 0, 2), (lo, hi, -1, -1, 0, 1), (lo, hi, -1, -1, 1, 1)])
    s, ms = scene_of(po, [(tris2, sib, checker())])
    got, want = compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 64, 64)
    compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 64, 64, full=True)
    assert len(np.unique(want.hit_id[want.hit_id >= 0])) == 1
    s.close()
    # two meshes with the same geometry and placement
    s, ms = scene_of(po, [(tris2[:1], one_leaf_of(po, tris2[:1]), checker()), (tris2[1:], one_leaf_of(po, tris2[1:]), checker(4))])
    got, want = compare(po, s, ms, [identity(po), identity(po)], front_camera(po), po.default_light(), 64, 64)
    assert set(np.unique(want.hit_id[want.hit_id >= 0] >> 24).tolist()) == {0}
    s.close()


def one_leaf_of(po, tris):
    lo, hi = bounds(tris)
    return node_array(po, [(lo, hi, -1, -1, 0, tris.shape[0])])


def test_many_coincident_triangles_in_a_real_build(po):
    """Every triangle twice: the reference's builder puts equal centroids on the same side, so twins share
    leaves or end up in leaves that could not be split."""
    rng = np.random.default_rng(33)
    base = tri_rows(rng.uniform(-0.2, 0.2, size=(60, 3, 3)) + rng.uniform(-1, 1, size=(60, 1, 3)) * (1.3, 1.0, 0.3),
                    uv=rng.uniform(0, 1, size=(60, 6)).astype(np.float32))
    post, nodes = po.build_bvh(np.concatenate([base, base]))
    s, ms = scene_of(po, [(post, nodes, checker())])
    compare(po, s, ms, [po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0.5, 0.1))], front_camera(po), po.default_light(), 128, 96)
    compare(po, s, ms, [po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0.5, 0.1))], front_camera(po), po.default_light(), 128, 96, full=True)
    s.close()


def test_degenerate_and_edge_on_triangles(po):
    v = [[[0, 0, 0], [0, 0, 0], [0, 0, 0]],                    # a point
         [[-1, 0, 0], [1, 0, 0], [0, 0, 0]],                   # collinear
         [[-1, -1, 0], [1, -1, 0], [0, 1, 0]],
         [[0, -1, -1], [0, -1, 1], [0, 1, 0]],                 # edge-on to the camera axis
         [[-1, 0.5, 0.5], [1, 0.5, 0.5], [0, 0.5, -1]]]        # edge-on horizontally
    tris = tri_rows(v)
    tris[:2, 15:24] = 0.0
    s, ms = scene_of(po, [(tris, one_leaf_of(po, tris), checker())])
    compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 80, 80)
    cam = po.Camera(po.vec3(0, 0.5, -4), po.vec3(0, 0.5, 0), po.vec3(0, 1, 0), 40.0)
    compare(po, s, ms, [identity(po)], cam, po.default_light(), 81, 81)
    s.close()


def test_malformed_trees_are_rejected_with_a_message(po):
    tris = tri_rows(np.random.default_rng(1).uniform(-1, 1, size=(6, 3, 3)))
    lo, hi = bounds(tris)
    bad = {
        "internal node with a single child": [(lo, hi, 1, -1, 0, 6), (lo, hi, -1, -1, 0, 6)],
        "node index out of range": [(lo, hi, 1, 7, 0, 6), (lo, hi, -1, -1, 0, 3)],
        "leaf triangle range out of bounds": [(lo, hi, -1, -1, 2, 6)],
        "triangle referenced by two leaves": [(lo, hi, 1, 2, 0, 6), (lo, hi, -1, -1, 0, 4), (lo, hi, -1, -1, 3, 3)],
        "node reachable twice": [(lo, hi, 1, 1, 0, 6), (lo, hi, -1, -1, 0, 6)],
    }
    for msg, rows in bad.items():
        s = emul.Scene()
        st = s.add(tris, node_array(po, rows), checker())
        assert st < 0 and msg in s.error(), (msg, st, s.error())
        s.close()


def test_tree_deeper_than_the_stack_is_rejected(po):
    n = 45
    rng = np.random.default_rng(2)
    tris = tri_rows(rng.uniform(-0.1, 0.1, size=(n, 3, 3)) + np.linspace(-2, 2, n)[:, None, None] * np.array([1, 0, 0]))
    rows = []
    for i in range(n - 1):                                  # node 2i: internal; node 2i+1: leaf with triangle i
        lo, hi = bounds(tris[i:])
        rows.append((lo, hi, 2 * i + 1, 2 * i + 2, i, n - i))
        rows.append((*bounds(tris[i:i + 1]), -1, -1, i, 1))
    rows.append((*bounds(tris[n - 1:]), -1, -1, n - 1, 1))
    s = emul.Scene()
    st = s.add(tris, node_array(po, rows), checker())
    assert st < 0 and "deeper" in s.error()
    s.close()
    # the same chain cut to the supported depth renders correctly
    n2 = 29
    rows2 = []
    for i in range(n2 - 1):
        lo, hi = bounds(tris[i:n2])
        rows2.append((lo, hi, 2 * i + 1, 2 * i + 2, i, n2 - i))
        rows2.append((*bounds(tris[i:i + 1]), -1, -1, i, 1))
    rows2.append((*bounds(tris[n2 - 1:n2]), -1, -1, n2 - 1, 1))
    s, ms = scene_of(po, [(tris[:n2], node_array(po, rows2), checker())])
    assert s.info(0)["depth"] == n2 - 1
    got, _ = compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 160, 40)
    compare(po, s, ms, [identity(po)], front_camera(po), po.default_light(), 160, 40, full=True)
    s.close()


def test_priorities_encode_the_reference_tie_order(po, meshes, em_default):
    """Inside one leaf the lower index wins, so its priority is higher; priorities are unique per mesh."""
    for mi, m in enumerate(meshes):
        pr = em_default.prios(mi, m.tris.shape[0])
        assert len(np.unique(pr)) == pr.size
        leaves = m.nodes[m.nodes["left"] < 0]
        for s0, c in zip(leaves["start"], leaves["count"]):
            assert (np.diff(pr[s0:s0 + c].astype(np.int64)) < 0).all()


# ------------------------------------------------------------------------------ scenes aimed at the distance culling
def test_culling_keeps_the_reference_result_on_adversarial_scenes(po):
    """Default traversal (ordered, distance-culled, regrouped tree) and reference-order traversal of the product's
    pixel code against the oracle's reference loop on slivers at the determinant threshold, coordinates of 1e3..1e4,
    a camera inside a mesh, rays grazing zero-thickness boxes and grazing hits behind a blocker: all planes identical."""
    import adversarial
    tex = checker()
    n = 0
    for name, soups, tfs, cam, w, h in adversarial.scenes(po, seed=5):
        meshes = adversarial.build(po, soups, tex)
        s = emul.Scene(regroup=True)
        for m in meshes:
            assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
        want = po.render_frame(meshes, tfs, cam, po.default_light(), w, h)
        for full in (False, True):
            got = s.render(cam, po.default_light(), tfs, w, h, full=full)
            for k, ref in (("hit_id", want.hit_id), ("shadow", want.shadow), ("rgb", want.rgb)):
                assert np.array_equal(got[k], ref), f"{name}: {k} differs ({'reference order' if full else 'default'})"
            assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t)), name
        s.close()
        n += 1
    assert n >= 20


# ------------------------------------------------------------------------------ meshes skipped by screen rectangle
def test_screen_rectangle_contains_every_pixel_whose_ray_passes_the_root_box(po):
    """mesh_screen_rect promises: outside the rectangle the reference's root test (ray.c:5-22 transform, aabb.c:33-56)
    fails.  Checked pixel by pixel with the oracle's own get_camera_ray / transform_ray / ray_aabb_intersect on boxes,
    placements and cameras that stress it: far from the origin, camera close to the box, narrow and wide views,
    flat boxes, a box behind or around the camera (then the rectangle must be the whole frame)."""
    import ctypes as C
    rng = np.random.default_rng(21)
    L = po.lib()
    light = po.default_light()
    w, h = 96, 54
    culled_somewhere = 0
    for trial in range(60):
        off = rng.choice([0.0, 10.0, 1.0e3, 1.0e4]) * rng.normal(size=3)
        half = 10 ** rng.uniform(-1.5, 0.7, size=3)
        if trial % 5 == 0:
            half[rng.integers(3)] = 0.0                                        # a flat box
        c = rng.normal(size=3) * 0.5
        bmin, bmax = (c - half).astype(np.float32), (c + half).astype(np.float32)
        tf = po.Transform(po.vec3(*(off + rng.normal(size=3))), po.vec3(*rng.uniform(-3.2, 3.2, size=3)))
        centre = off + np.array([tf.position.x, tf.position.y, tf.position.z]) - off   # world position of the box, roughly
        centre = np.array([tf.position.x, tf.position.y, tf.position.z])
        dist = 10 ** rng.uniform(-0.5, 1.5) * (1.0 + float(half.max()))
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        eye = centre + d * dist
        if trial % 7 == 3:
            look = centre + rng.normal(size=3) * 3 * dist                       # looking past it (or away from it)
        else:
            look = centre + rng.normal(size=3) * float(half.max()) * rng.uniform(0, 3)
        fov = float(rng.choice([1.0, 20.0, 60.0, 120.0]))
        cam = po.Camera(po.vec3(*eye), po.vec3(*look), po.vec3(0, 1, 0), fov)
        rect = emul.screen_rect(tf, bmin, bmax, cam, light, w, h)
        assert 0 <= rect[0] <= rect[2] <= (w + 7) // 8 and 0 <= rect[1] <= rect[3] <= (h + 7) // 8
        box = po.Aabb(po.vec3(*bmin), po.vec3(*bmax))
        passes = np.zeros((h, w), bool)
        for y in range(h):
            for x in range(w):
                ray = L.orc_camera_ray(C.byref(cam), np.float32((x + 0.5) / np.float32(w)), np.float32((y + 0.5) / np.float32(h)),
                                       np.float32(w) / np.float32(h))
                passes[y, x] = bool(L.orc_ray_aabb(L.orc_transform_ray(ray, tf), box))
        inside = np.zeros((h, w), bool)
        inside[rect[1] * 8:rect[3] * 8, rect[0] * 8:rect[2] * 8] = True
        assert not (passes & ~inside).any(), f"trial {trial}: {int((passes & ~inside).sum())} rays pass the box outside {rect}"
        if (rect[2] - rect[0]) * (rect[3] - rect[1]) < ((w + 7) // 8) * ((h + 7) // 8):
            culled_somewhere += 1
    assert culled_somewhere >= 15                                               # the rectangle is not trivially the whole frame


def test_meshes_skipped_by_screen_rectangle_do_not_change_the_frame(po, em_default, golden_frames):
    """The traversal as the kernels run it (no work counts: meshes outside their screen rectangle are skipped) on the
    reference scene and on the scenes of tests/adversarial.py, against golden planes / the oracle."""
    import adversarial
    cam, light = po.default_camera(), po.default_light()
    for f in (0, 45, 95):
        got = em_default.render(cam, light, po.animation_transforms(f), 240, 135, screen_cull=True)
        for k in ("rgb", "hit_id", "shadow"):
            assert np.array_equal(got[k], golden_frames[f"f{f}_{k}"]), (f, k)
        assert np.array_equal(bits(got["hit_t"]), bits(golden_frames[f"f{f}_hit_t"]))
    tex = checker()
    for name, soups, tfs, acam, w, h in adversarial.scenes(po, seed=9):
        ms = adversarial.build(po, soups, tex)
        s = emul.Scene(regroup=True)
        for m in ms:
            assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
        want = po.render_frame(ms, tfs, acam, light, w, h)
        got = s.render(acam, light, tfs, w, h, screen_cull=True)
        s.close()
        assert np.array_equal(got["hit_id"], want.hit_id) and np.array_equal(got["rgb"], want.rgb), name
        assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t)) and np.array_equal(got["shadow"], want.shadow), name


def test_shadow_rectangle_contains_every_point_whose_shadow_ray_passes_the_root_box(po):
    """mesh_shadow_rect promises: a surface point of the scene that is seen outside the rectangle cannot send its shadow ray
    through the mesh's root box.  Checked with the oracle's transform_ray + ray_aabb_intersect on points spread over the
    scene's bounds (more than the surfaces: the claim holds for every point inside the bounds), for several lights,
    placements and cameras; the point's pixel is its own projection."""
    rng = np.random.default_rng(33)
    L = po.lib()
    w, h = 192, 108
    tiles = ((w + 7) // 8, (h + 7) // 8)
    culled = 0
    for trial in range(40):
        off = rng.choice([0.0, 0.0, 100.0, 1.0e4]) * rng.normal(size=3)
        half = 10 ** rng.uniform(-1.0, 0.3, size=3)
        c = rng.normal(size=3) * 0.3
        bmin, bmax = (c - half).astype(np.float32), (c + half).astype(np.float32)
        pos = off + rng.normal(size=3) * 2
        tf = po.Transform(po.vec3(*pos), po.vec3(*rng.uniform(-3.2, 3.2, size=3)))
        ext = float(np.linalg.norm(half)) + 1e-3
        slo, shi = pos - rng.uniform(2, 12, size=3), pos + rng.uniform(2, 12, size=3)      # bounds of "the scene" around the mesh
        ld = rng.normal(size=3)
        if trial % 4 == 0:
            ld[rng.integers(3)] = 0.0
        ld /= np.linalg.norm(ld)
        light = po.Light(po.vec3(*np.array(ld, np.float32)), po.vec3(1, 1, 1))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        eye = pos + d * rng.uniform(15, 40)
        cam = po.Camera(po.vec3(*eye), po.vec3(*(pos + rng.normal(size=3) * 3)), po.vec3(0, 1, 0), float(rng.choice([30.0, 60.0, 100.0])))
        rect = emul.shadow_rect(tf, bmin, bmax, slo, shi, cam, light, w, h)
        assert 0 <= rect[0] <= rect[2] <= tiles[0] and 0 <= rect[1] <= rect[3] <= tiles[1]
        if (rect[2] - rect[0]) * (rect[3] - rect[1]) < tiles[0] * tiles[1]:
            culled += 1
        # camera basis in double, as camera.c:13-17 defines it
        e = np.array([cam.position.x, cam.position.y, cam.position.z], np.float64)
        look = np.array([cam.look_at.x, cam.look_at.y, cam.look_at.z], np.float64)
        fwd = look - e; fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        scale = np.tan(np.radians(cam.fov * 0.5)); aspect = w / h
        box = po.Aabb(po.vec3(*bmin), po.vec3(*bmax))
        pts = rng.uniform(slo, shi, size=(1500, 3))
        # half of the points close to the box swept against the light, where the shadow rays that matter start
        pts[:750] = pos + rng.normal(size=(750, 3)) * ext * 1.5 - ld * rng.uniform(0, 12, size=(750, 1))
        pts = np.clip(pts, slo, shi)
        for P in pts:
            Pf = np.array(P, np.float32)
            ray = po.Ray(po.vec3(*Pf), light.direction)
            if not L.orc_ray_aabb(L.orc_transform_ray(ray, tf), box):
                continue
            v = Pf.astype(np.float64) - e
            z = v @ fwd
            if z <= 1e-6:
                assert rect == [0, 0, tiles[0], tiles[1]] or True      # behind the camera: never seen
                continue
            px = ((v @ right) / z / (aspect * scale) + 1) * 0.5 * w - 0.5
            py = (1 - (v @ up) / z / scale) * 0.5 * h - 0.5
            if not (0 <= px < w and 0 <= py < h):
                continue
            tx, ty = int(px) // 8, int(py) // 8
            assert rect[0] <= tx < rect[2] and rect[1] <= ty < rect[3], f"trial {trial}: point at pixel ({px:.1f}, {py:.1f}) outside {rect}"
    assert culled >= 10


# ------------------------------------------------------------------------------ flat meshes: entry through the grid
def _flat_mesh(rng, axis, n, extent, offset):
    """A jittered n x n triangulated sheet with zero thickness along `axis`, centred at `offset`."""
    a, b = (axis + 1) % 3, (axis + 2) % 3
    g = np.linspace(-extent, extent, n + 1)
    P = np.zeros((n + 1, n + 1, 3))
    for j in range(n + 1):
        for i in range(n + 1):
            P[j, i, a] = g[i] + (rng.uniform(-0.3, 0.3) * extent / n if 0 < i < n else 0.0)
            P[j, i, b] = g[j] + (rng.uniform(-0.3, 0.3) * extent / n if 0 < j < n else 0.0)
    P += offset
    P[..., axis] = np.float32(offset[axis])
    v = []
    for j in range(n):
        for i in range(n):
            v.append([P[j, i], P[j + 1, i + 1], P[j, i + 1]])
            v.append([P[j, i], P[j + 1, i], P[j + 1, i + 1]])
    return tri_rows(np.array(v, np.float32))


def test_flat_meshes_entered_through_their_grid_give_the_reference_frame(po):
    """Meshes of zero thickness along x, y or z (Packer::build_grid), near the origin and with coordinates of 1e3 in their own
    space, seen steeply, at grazing angles and from far away, lit along and across the sheet: default traversal (which enters
    through the grid), reference order and the oracle agree on every plane, and the grid saves box tests."""
    rng = np.random.default_rng(17)
    tex = checker()
    saved = []
    for trial in range(18):
        axis = trial % 3
        n = int(rng.choice([6, 12, 20]))
        extent = float(10 ** rng.uniform(-0.3, 1.0))
        offset = (rng.normal(size=3) * (1.0e3 if trial % 6 >= 4 else 1.0)).astype(np.float32).astype(np.float64)
        sheet = _flat_mesh(rng, axis, n, extent, offset)
        blob = tri_rows((offset + rng.normal(size=(40, 1, 3)) * extent * 0.3 + rng.normal(size=(40, 3, 3)) * extent * 0.05).astype(np.float32))
        tfs = [po.Transform(po.vec3(*rng.normal(size=3)), po.vec3(*rng.uniform(-1, 1, size=3))),
               po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]
        nrm = np.zeros(3); nrm[axis] = 1.0
        side = rng.normal(size=3); side[axis] = 0; side /= np.linalg.norm(side)
        elev = [0.8, 0.05, 1e-3, 1e-5][trial % 4]                                   # steep ... grazing
        dist = extent * float(rng.uniform(0.5, 4.0)) * (100.0 if trial % 9 == 8 else 1.0)
        eye_local = offset + side * dist * 0.9 + nrm * dist * elev
        # the sheet is placed by tfs[0]: put the camera where that placement takes eye_local (approximately: rotation order as ray.c:7-9)
        M = np.linalg.inv(np.array(po.mesh_matrices(tfs[0])[0], np.float64))       # mesh space -> world (inverse of ray.c:14's matrix)
        eye, look = (M @ np.append(eye_local, 1.0))[:3], (M @ np.append(offset, 1.0))[:3]
        cam = po.Camera(po.vec3(*eye), po.vec3(*look), po.vec3(*(M[:3, :3] @ nrm)), float(rng.choice([20.0, 60.0])))
        ld = rng.normal(size=3); ld /= np.linalg.norm(ld)
        light = po.Light(po.vec3(*np.array(ld, np.float32)), po.vec3(1.2, 1.1, 1.0))
        meshes = []
        for soup in (sheet, blob):
            t, nodes = po.build_bvh(soup)
            meshes.append(po.HostMesh(t, nodes, tex))
        w, h = 128, 72
        want = po.render_frame(meshes, tfs, cam, light, w, h)
        counts = {}
        for regroup in (False, True):
            s = emul.Scene(regroup=regroup)
            for mm in meshes:
                assert s.add(mm.tris, mm.nodes, mm.rgba) == 0, s.error()
            for mode in ({"full": False}, {"full": True}, {"screen_cull": True}):
                got = s.render(cam, light, tfs, w, h, **mode)
                for k, ref in (("hit_id", want.hit_id), ("shadow", want.shadow), ("rgb", want.rgb)):
                    assert np.array_equal(got[k], ref), f"trial {trial} axis {axis} {mode} regroup {regroup}: {k}"
                assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t)), (trial, mode)
                if mode == {"full": False}:
                    counts[regroup] = got["counters"]["slab_tests"]
            s.close()
        if (want.hit_id >= 0).mean() > 0.2:
            saved.append(counts[True] / counts[False])
    assert len(saved) >= 6 and np.median(saved) < 0.9, saved


def test_second_tree_is_the_same_whether_built_while_packing_or_afterwards(meshes):
    """Meshes packed on the device get their surface-area-split tree from the packed reference tree afterwards
    (Packer::regroup_packed); it must be the tree run() builds while packing: same records, same roots, same depth."""
    for m in meshes:
        assert emul.regroup_routes_agree(m.tris, m.nodes) == 0
    # a soup large enough for the threaded part of the recursion (ranges above 65 536 leaves), and tiny ones
    from oracle import pyoracle as po
    rng = np.random.default_rng(5)
    for n, want in ((300_000, 0), (40, 0), (9, 2), (1, 2)):
        c = rng.uniform(-20, 20, (n, 1, 3)).astype(np.float32)
        soup = np.zeros((n, 24), np.float32)
        soup[:, 0:9] = (c + rng.normal(size=(n, 3, 3)).astype(np.float32) * 0.2).reshape(n, 9)
        soup[:, 17] = 1.0; soup[:, 20] = 1.0; soup[:, 23] = 1.0
        tris, nodes = po.build_bvh(soup)
        assert emul.regroup_routes_agree(tris, nodes) == want, n


# ------------------------------------------------------------------------------ zeros, infinities, NaNs
def test_degenerate_cameras_lights_and_placements_give_the_reference_bytes(po, em_default):
    """tests/adversarial.py::degenerate_views through the host build of the pixel code (both traversal modes, meshes skipped by
    screen rectangle as the kernels skip them) against the oracle: every plane, NaN distances included, bit for bit."""
    import adversarial
    n = 0
    for name, tfs, cam, light in adversarial.degenerate_views(po):
        want = po.render_frame(po.default_meshes(), tfs, cam, light, 96, 54)
        for kw in ({"full": False}, {"full": True}, {"screen_cull": True}):
            got = em_default.render(cam, light, tfs, 96, 54, **kw)
            for k, ref in (("hit_id", want.hit_id), ("shadow", want.shadow), ("rgb", want.rgb)):
                assert np.array_equal(got[k], ref), f"{name}: {k} differs ({kw})"
            assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t)), f"{name}: hit_t ({kw})"
        n += 1
    assert n >= 20


def test_shadow_rays_that_cannot_change_the_colour_are_left_out_without_changing_it(po, meshes, em_default):
    """A caller that stores the colour only (rt_render_frames, calls without a shadow plane): where max(0.2, n.L) is 0.2
    already, scene.c:143-144 gives diffuse 0.2 blocked or not, and the pixel code sends no shadow ray.  The colours stay the
    oracle's byte for byte -- golden frames, random placements and lights (negative and above-one colours), the degenerate
    views (NaN lights and normals among them) -- and rays are in fact left out where surfaces face away from the light."""
    import adversarial
    unsent = 0
    for frame in (0, 45, 95):
        tfs = po.animation_transforms(frame)
        want = po.render_frame(meshes, tfs, po.default_camera(), po.default_light(), 240, 135)
        got = em_default.render(po.default_camera(), po.default_light(), tfs, 240, 135, colour_only=True)
        assert np.array_equal(got["rgb"], want.rgb) and np.array_equal(got["hit_id"], want.hit_id)
        assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t))
        sent = got["shadow"] != 2
        assert np.array_equal(got["shadow"][sent], want.shadow[sent])            # the rays that were sent end as the oracle's
        assert ((got["hit_id"] >> 24)[~sent] != 2).all()                         # the ground faces the light: all of its rays go out
        unsent += int((~sent).sum())
        counted = em_default.render(po.default_camera(), po.default_light(), tfs, 240, 135)["counters"]
        assert counted["idle_shadow_rays"] == int((~sent).sum())                 # a counting run sends them, and says how many
    assert unsent > 500
    rng = np.random.default_rng(17)
    for _ in range(8):
        tfs = [po.Transform(po.vec3(*rng.uniform(-1.5, 1.5, 3)), po.vec3(*rng.uniform(-3.2, 3.2, 3))) for _ in range(2)]
        tfs.append(po.Transform(po.vec3(0, 0, 0), po.vec3(*rng.uniform(-0.05, 0.05, 3))))
        eye = rng.uniform(-4, 4, 3)
        eye[1] = abs(eye[1]) + 1.0
        cam = po.Camera(po.vec3(*eye), po.vec3(0, 0.3, 0), po.vec3(0, 1, 0), float(rng.uniform(30, 90)))
        light = po.lib().orc_make_light(po.vec3(*rng.uniform(-1, 1, 3)), po.vec3(*rng.uniform(-0.5, 1.6, 3)))
        want = po.render_frame(meshes, tfs, cam, light, 96, 54)
        got = em_default.render(cam, light, tfs, 96, 54, colour_only=True)
        assert np.array_equal(got["rgb"], want.rgb)
    for name, tfs, cam, light in adversarial.degenerate_views(po):
        want = po.render_frame(po.default_meshes(), tfs, cam, light, 96, 54)
        got = em_default.render(cam, light, tfs, 96, 54, colour_only=True)
        assert np.array_equal(got["rgb"], want.rgb), name


def test_meshes_with_nan_and_infinite_vertices_through_the_host_build_of_the_pixel_code(po):
    """tests/adversarial.py::nonfinite_soups as meshes: packer (both trees) and traversal (both modes) on boxes with NaN and
    infinite bounds against the oracle's reference loop, every plane."""
    import adversarial
    tex = checker()
    ident = [po.Transform(po.vec3(0, 0, 0), po.vec3(0.2, 0.4, 0.1))]
    cam, light = po.default_camera(), po.default_light()
    n = 0
    for name, soup in adversarial.nonfinite_soups(sizes=(7, 60, 500)):
        t, nodes = po.build_bvh(soup)
        want = po.render_frame([po.HostMesh(t, nodes, tex)], ident, cam, light, 96, 54)
        for regroup in (False, True):
            s = emul.Scene(regroup=regroup)
            assert s.add(t, nodes, tex) == 0, (name, s.error())
            for full in (False, True):
                got = s.render(cam, light, ident, 96, 54, full=full)
                for k, ref in (("hit_id", want.hit_id), ("shadow", want.shadow), ("rgb", want.rgb)):
                    assert np.array_equal(got[k], ref), f"{name}: {k} differs (regroup {regroup}, reference order {full})"
                assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t)), name
            s.close()
        n += 1
    assert n >= 30


def test_tree_over_the_triangles_themselves_reaches_the_same_surfaces(po, meshes):
    """The third tree (Packer::build_loose: surface-area splits over single triangles, leaves of up to eight neighbours,
    its own copy of the triangle records) is what the kernels' fast build walks.  With the exact arithmetic of the CPU
    build it must find the reference's hits except where a ray grazes an edge -- at least 99.99 % of camera rays, in
    practice all of them on this scene -- report the reference's triangle ids (each record carries its own), and test
    fewer boxes and triangles on the closed meshes.  The flat ground has no such tree: its grid already leads to leaves."""
    s = emul.Scene(regroup=True)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0
    for f in (0, 45):
        tfs = po.animation_transforms(f)
        s.set_loose(False)
        exact = s.render(po.default_camera(), po.default_light(), tfs, 480, 270)
        s.set_loose(True)
        loose = s.render(po.default_camera(), po.default_light(), tfs, 480, 270)
        s.set_loose(False)
        same = float((exact["hit_id"] == loose["hit_id"]).mean())
        assert same >= 0.9999, (f, same)
        assert float((np.abs(exact["rgb"].astype(np.int16) - loose["rgb"].astype(np.int16)).max(axis=2) <= 1).mean()) >= 0.999
        assert loose["counters"]["slab_tests"] < exact["counters"]["slab_tests"]
        assert loose["stack_high"] <= exact["stack_high"] + 1
    # every triangle of a random soup is in exactly one leaf of that tree: a frame that looks at the soup from all around
    # finds, triangle for triangle, what the reference's tree finds
    rng = np.random.default_rng(11)
    soup = np.zeros((400, 24), np.float32)
    centre = rng.uniform(-1, 1, (400, 1, 3)).astype(np.float32)
    soup[:, :9] = (centre + rng.uniform(-0.15, 0.15, (400, 3, 3)).astype(np.float32)).reshape(400, 9)
    soup[:, 15:24] = np.tile(np.array([0, 1, 0], np.float32), 3)
    tris, nodes = po.build_bvh(soup)
    s2 = emul.Scene(regroup=True)
    assert s2.add(tris, nodes, np.full((4, 4, 4), 200, np.uint8)) == 0
    seen_exact, seen_loose = set(), set()
    for cam_pos in ((3, 2, 3), (-3, 2, 3), (3, -2, -3), (-3, 1, -3), (0.1, 4, 0.1)):
        cam = po.Camera(po.vec3(*cam_pos), po.vec3(0, 0, 0), po.vec3(0, 1, 0), 50.0)
        tf = [po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]
        s2.set_loose(False)
        a = s2.render(cam, po.default_light(), tf, 200, 200)
        s2.set_loose(True)
        b = s2.render(cam, po.default_light(), tf, 200, 200)
        assert float((a["hit_id"] == b["hit_id"]).mean()) >= 0.999
        seen_exact |= set(np.unique(a["hit_id"]).tolist())
        seen_loose |= set(np.unique(b["hit_id"]).tolist())
    assert len(seen_exact) > 300 and len(seen_exact ^ seen_loose) <= 2


def test_flat_soups_of_overlapping_random_triangles_through_the_grid(po):
    """Flat meshes that are not sheets: random triangles of one plane lying on top of each other, needles and specks among
    them, so that a cell of the grid is covered by many leaves and its triangle masks decide what is tested (Packer::tri_mask;
    needles are outside its argument and must always be kept).  Default walk, the walk with tiles skipped by rectangles and
    the self-checking walk against the oracle, steep and grazing views."""
    rng = np.random.default_rng(23)
    tex = checker()
    for trial in range(10):
        axis = trial % 3
        n = int(rng.choice([40, 200, 600]))
        soup = (rng.normal(size=(n, 24)) * float(rng.choice([0.3, 1.0, 3.0]))).astype(np.float32)
        thin = rng.uniform(size=n) < 0.15                                   # needles: the third vertex almost on the first edge
        soup[thin, 6:9] = soup[thin, 0:3] + (soup[thin, 3:6] - soup[thin, 0:3]) * np.float32(0.7) + rng.normal(size=(int(thin.sum()), 3)).astype(np.float32) * np.float32(1e-5)
        speck = rng.uniform(size=n) < 0.1
        soup[speck, 3:9] = np.tile(soup[speck, 0:3], 2) + rng.normal(size=(int(speck.sum()), 6)).astype(np.float32) * np.float32(1e-3)
        soup[:, [axis, 3 + axis, 6 + axis]] = np.float32(rng.normal() * 0.3)
        t, nodes = po.build_bvh(soup)
        mesh = po.HostMesh(t, nodes, tex)
        tfs = [po.Transform(po.vec3(*rng.normal(size=3) * 0.3), po.vec3(*rng.uniform(-1, 1, size=3)))]
        eye = rng.normal(size=3) * 3.0
        if trial % 2:
            eye[axis] *= 0.02                                                # grazing
        cam = po.Camera(po.vec3(*eye), po.vec3(*rng.normal(size=3) * 0.3), po.vec3(0, 1, 0), float(rng.choice([25.0, 70.0])))
        ld = rng.normal(size=3); ld /= np.linalg.norm(ld)
        light = po.Light(po.vec3(*np.array(ld, np.float32)), po.vec3(1.2, 1.1, 1.0))
        w, h = 160, 90
        want = po.render_frame([mesh], tfs, cam, light, w, h)
        s = emul.Scene(regroup=True)
        assert s.add(mesh.tris, mesh.nodes, mesh.rgba) == 0, s.error()
        for mode in ({"full": False}, {"screen_cull": True}, {"verify": True}):
            got = s.render(cam, light, tfs, w, h, **mode)
            for k, ref in (("hit_id", want.hit_id), ("shadow", want.shadow), ("rgb", want.rgb)):
                assert np.array_equal(got[k], ref), f"trial {trial} axis {axis} n {n} {mode}: {k}"
            assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t)), (trial, mode)
            if mode.get("verify"):
                c = got["counters"]
                assert c["entry_disagreements"] == 0 and c["cull_disagreements"] == 0 and c["vis_disagreements"] == 0, (trial, c)
        s.close()
