"""Scenes with many placed meshes.  The reference's Scene.meshes may hold the same geometry under any
number of transforms (add_mesh_to_scene, scene.c:31-35) and render_scene tests every ray against every
entry (scene.c:72-108, 124-139); the oracle does exactly that.  The product proposes instances through a
hierarchy over their world boxes and must produce the same planes, bit for bit."""
import numpy as np
import pytest

import emul
from conftest import bits


def grid_scene(po, n_side, seed, frame=10, spacing=1.1):
    """drone, ground and an n_side x n_side grid of treasures with seeded heights and rotations.
    Returns (mesh ids, transforms)."""
    rng = np.random.default_rng(seed)
    ids = [0, 2]
    tfs = [po.animation_transforms(frame)[0], po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]
    half = (n_side - 1) / 2.0
    for gx in range(n_side):
        for gz in range(n_side):
            ids.append(1)
            tfs.append(po.Transform(po.vec3((gx - half) * spacing, 0.45 + 0.1 * rng.uniform(), (gz - half) * spacing),
                                    po.vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
    return ids, tfs


def oracle_planes(po, meshes, ids, tfs, cam, light, w, h):
    fr, inst, tri = po.render_frame_ids([meshes[i] for i in ids], tfs, cam, light, w, h)     # list index and triangle as two planes
    return fr, inst, tri


@pytest.fixture(scope="module")
def em_default(meshes):
    s = emul.Scene()
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    yield s
    s.close()


def check(got, fr, inst, tri):
    assert np.array_equal(got["hit_inst"], inst), int((got["hit_inst"] != inst).sum())
    assert np.array_equal(got["hit_tri"], tri)
    assert np.array_equal(bits(got["hit_t"]), bits(fr.hit_t))
    assert np.array_equal(got["shadow"], fr.shadow)
    assert np.array_equal(got["rgb"], fr.rgb)


# ------------------------------------------------------------------------------------------ CPU

def test_three_instances_are_the_reference_scene(po, meshes, em_default, golden_frames):
    """One instance per mesh in the driver's order = the driver's scene."""
    for f in (0, 45, 95):
        got = em_default.render_instanced(po.default_camera(), po.default_light(), [0, 1, 2], po.animation_transforms(f), 240, 135)
        assert np.array_equal(got["rgb"], golden_frames[f"f{f}_rgb"])
        want = golden_frames[f"f{f}_hit_id"]
        assert np.array_equal(np.where(got["hit_inst"] >= 0, (got["hit_inst"] << 24) | got["hit_tri"], -1), want)
        assert np.array_equal(bits(got["hit_t"]), bits(golden_frames[f"f{f}_hit_t"]))
        assert np.array_equal(got["shadow"], golden_frames[f"f{f}_shadow"])


@pytest.mark.parametrize("n_side,size", [(1, (96, 54)), (5, (160, 90)), (10, (120, 68))])
def test_grid_of_instances_equals_reference_loop(po, meshes, em_default, n_side, size):
    ids, tfs = grid_scene(po, n_side, seed=n_side)
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), *size)
    got = em_default.render_instanced(po.default_camera(), po.default_light(), ids, tfs, *size)
    check(got, fr, inst, tri)
    assert got["counters"]["hit_pixels"] == fr.counters["hit_pixels"] and got["counters"]["shadowed_pixels"] == fr.counters["shadowed_pixels"]
    # the hierarchy spares most of the reference's per-mesh work
    if n_side >= 5:
        assert got["counters"]["slab_tests"] < fr.counters["slab_tests"] // 2
        assert (inst > 1).any() and (fr.shadow == 1).any()


def test_coincident_instances_earlier_list_entry_wins(po, meshes, em_default):
    """The same mesh twice at the same place: every hit distance ties; the reference keeps the earlier mesh (scene.c:82)."""
    tf = po.Transform(po.vec3(0.3, 0.5, 0.2), po.vec3(0.1, 0.7, 0))
    ground = po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))
    for ids, tfs, later in (([1, 1, 2], [tf, tf, ground], {1}), ([2, 1, 2, 1], [ground, tf, ground, tf], {2, 3})):
        fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), 120, 68)
        got = em_default.render_instanced(po.default_camera(), po.default_light(), ids, tfs, 120, 68)
        check(got, fr, inst, tri)
        hit = set(np.unique(inst[inst >= 0]).tolist())
        assert hit and not (hit & later)                       # never the later twin


def test_single_leaf_meshes_and_empty_meshes_as_instances(po, meshes):
    """Meshes whose BVH is one leaf use the caller's bound inside the leaf (bvh.c:85); empty meshes are never hit."""
    from test_host_emul import tri_rows, checker
    small = tri_rows([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[-1, -1, 0.5], [1, -1, 0.5], [0, 1, 0.5]]])
    post, nodes = po.build_bvh(small)
    e_post, e_nodes = po.build_bvh(np.zeros((0, 24), np.float32))
    s = emul.Scene()
    parts = [(post, nodes, checker()), (e_post, e_nodes, checker()), (meshes[2].tris, meshes[2].nodes, meshes[2].rgba)]
    host_meshes = []
    for t, n, r in parts:
        assert s.add(t, n, r) == 0
        host_meshes.append(po.HostMesh(np.ascontiguousarray(t, np.float32), n, r))
    ids = [0, 1, 0, 2, 0]
    tfs = [po.Transform(po.vec3(0, 1, 0), po.vec3(0, 0.3, 0)), po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0)),
           po.Transform(po.vec3(0.2, 1.1, 0.4), po.vec3(0.2, 0.3, 0)), po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0)),
           po.Transform(po.vec3(0, 1, 0), po.vec3(0, 0.3, 0))]                      # the last one coincides with the first
    fr, inst, tri = oracle_planes(po, host_meshes, ids, tfs, po.default_camera(), po.default_light(), 128, 72)
    got = s.render_instanced(po.default_camera(), po.default_light(), ids, tfs, 128, 72)
    check(got, fr, inst, tri)
    assert 0 in inst and 2 in inst and 4 not in inst and 1 not in inst
    s.close()


def test_thousands_of_instances_against_the_reference_loop(po, meshes, em_default):
    """3 138 list entries (about 9.9 million placed triangles) at 64x36: the reference's loop over all of them per ray
    is slow but still feasible at this size.  The oracle reports list index and triangle as two planes here (its packed
    id, mesh << 24 | triangle, wraps for long lists)."""
    ids, tfs = grid_scene(po, 56, seed=3, spacing=1.05)
    assert len(ids) == 56 * 56 + 2
    w, h = 64, 36
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), w, h)
    got = em_default.render_instanced(po.default_camera(), po.default_light(), ids, tfs, w, h)
    check(got, fr, inst, tri)
    assert inst.max() > 255
    assert got["tlas_depth"] <= 14 and len(np.unique(got["hit_inst"])) > 50


def test_kept_hierarchy_follows_a_changing_list(po, meshes):
    """The hierarchy over the instances is kept between frames and patched while few instances change: every frame
    of a sequence -- one mover (the drone), several movers, a mesh swapped, an instance emptied and filled again,
    nothing changed, most of the list moved (built anew), a shorter list -- equals the reference loop, and the kept
    tree was reused where the change was small."""
    s = emul.Scene()
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    e_post, e_nodes = po.build_bvh(np.zeros((0, 24), np.float32))
    assert s.add(e_post, e_nodes, meshes[0].rgba) == 0, s.error()                          # mesh 3: no triangles
    empty = po.HostMesh(np.ascontiguousarray(e_post, np.float32), e_nodes, meshes[0].rgba)
    all_meshes = list(meshes) + [empty]
    s.keep_hierarchy(True)
    ids, tfs = grid_scene(po, 6, seed=3)
    rng = np.random.default_rng(8)
    w, h = 128, 72
    cam, light = po.default_camera(), po.default_light()

    def frame(ids, tfs):
        fr, inst, tri = oracle_planes(po, all_meshes, ids, tfs, cam, light, w, h)
        got = s.render_instanced(cam, light, ids, tfs, w, h)
        check(got, fr, inst, tri)
        return got

    def somewhere():
        return po.Transform(po.vec3(*(rng.uniform(-2.5, 2.5, 3) * [1, 0.2, 1] + [0, 0.9, 0])), po.vec3(0, float(rng.uniform(0, 6)), 0))

    first = frame(ids, tfs)
    assert s.hierarchy_reused() == 0
    for f in (20, 30, 40):                                              # the drone flies: one mover per frame
        tfs[0] = po.animation_transforms(f)[0]
        got = frame(ids, tfs)
        assert got["tlas_pairs"] == first["tlas_pairs"] + 1             # the kept tree and a join above it
    assert s.hierarchy_reused() == 3
    for k in (4, 9, 17, 30):                                            # several movers, one of them twice
        tfs[k] = somewhere()
    frame(ids, tfs)
    tfs[9] = somewhere()
    ids[12] = 0                                                         # a treasure becomes a drone
    frame(ids, tfs)
    ids[6] = 3                                                          # an instance loses its triangles ...
    got = frame(ids, tfs)
    assert not (got["hit_inst"] == 6).any()
    ids[6] = 1                                                          # ... and gets them back
    got = frame(ids, tfs)
    assert (got["hit_inst"] == 6).any()
    frame(ids, tfs)                                                     # nothing changed
    assert s.hierarchy_reused() == 8
    for k in range(2, len(tfs), 2):                                     # half the list moves: built anew and kept
        tfs[k] = somewhere()
    frame(ids, tfs)
    assert s.hierarchy_reused() == 8
    tfs[0] = po.animation_transforms(50)[0]
    got = frame(ids, tfs)
    assert s.hierarchy_reused() == 9
    frame(ids[:11], tfs[:11])                                           # another length: built anew
    assert s.hierarchy_reused() == 9
    frame(ids[:11], tfs[:10] + [somewhere()])
    assert s.hierarchy_reused() == 10
    s.close()


def test_kept_hierarchy_stays_a_valid_tree_under_random_changes(tmp_path):
    """tests/host_emul/tlas_check.cpp, built with the address and undefined-behaviour sanitizers: after every step of
    random sequences (a few movers, emptied and refilled instances, many movers, other list lengths, no change) every
    instance of the list hangs in exactly one reachable leaf and every box contains what is below it."""
    import json
    import os
    import subprocess
    from conftest import REPO
    exe = str(tmp_path / "tlas_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-static-libasan", "-static-libubsan", "-o", exe, os.path.join(REPO, "tests", "host_emul", "tlas_check.cpp"), "-lm"], check=True, timeout=300)
    for seed in (1, 2, 3):
        r = subprocess.run([exe, str(seed), "400"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0"))
        assert r.returncode == 0, r.stderr[-2000:]
        info = json.loads(r.stdout.strip().splitlines()[-1])
        assert info["errors"] == 0 and info["reused"] > 200


def nonfinite_lists(po):
    """Yields (name, mesh ids, transforms): a 6 x 6 grid scene in which a tenth, half or all of the 38 list entries have NaN or
    infinite angles or positions.  The reference's loop just finds no hit in such an entry (its ray transform yields NaNs); the
    hierarchy over the world boxes has to cope with boxes that are NaN or infinite and still propose every other entry."""
    nan, inf = float("nan"), float("inf")
    rng = np.random.default_rng(4)
    kinds = (("NaN angles", lambda t: po.Transform(t.position, po.vec3(0, nan, 0))),
             ("NaN positions", lambda t: po.Transform(po.vec3(nan, 0, 0), t.rotation)),
             ("infinite positions", lambda t: po.Transform(po.vec3(inf, 0.5, -inf), t.rotation)),
             ("positions of 1e30", lambda t: po.Transform(po.vec3(1e30, 0.5, 1e30), t.rotation)),
             ("infinite angles", lambda t: po.Transform(t.position, po.vec3(inf, 0, 0))))
    for label, bad in kinds:
        for frac in (0.1, 0.5, 1.0):
            ids, tfs = grid_scene(po, 6, seed=3)
            for k in range(len(tfs)):
                if rng.uniform() < frac:
                    tfs[k] = bad(tfs[k])
            yield f"{label} on {frac:.0%} of the entries", ids, tfs


def test_entries_with_nan_and_infinite_placements(po, meshes, em_default):
    n = 0
    for name, ids, tfs in nonfinite_lists(po):
        fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), 96, 54)
        check(em_default.render_instanced(po.default_camera(), po.default_light(), ids, tfs, 96, 54), fr, inst, tri)
        n += 1
    assert n == 15


def far_entries_scene(po):
    """Entries placed at +-1e30 among ordinary ones, the light straight along z (two components of -0), found by
    scripts/soak_parity.py --instanced --wild: the reference's test in the space of an entry at 1e30 works on ray origins of that
    size, accepts and rejects by rounding alone, and its shadow rays -- which start ON such entries or head for one 1e30
    away -- do the same against every other entry.  The world boxes of the hierarchy have to allow for that."""
    P = [(0.04725063592195511, 180.0, 0.6031175255775452, 3.033522605895996, 0.7726259827613831, 0.7332237958908081),
         (0.6915498971939087, -0.23808303475379944, 0.10843290388584137, 2.77695369720459, -2.682859182357788, -2.1754519939422607),
         (-0.3740750849246979, -0.014469670131802559, 1.0, 180.0, -2.951401948928833, 2.185908317565918),
         (1.0000000150474662e+30, 0.021858487278223038, -1.0000000116860974e-07, 0.843559741973877, 1.5415972471237183, -0.29535576701164246),
         (0.6323668360710144, 0.12587839365005493, -0.7777173519134521, -0.9772530198097229, -0.21872852742671967, -0.01635465957224369),
         (-0.0, 1.0000000150474662e+30, -0.20492178201675415, -3.199599027633667, 0.4215320646762848, -1.234921932220459),
         (-0.7121354341506958, 0.2877514064311981, 1.0005271035279194e-42, -1.2818286418914795, 2.015014410018921, 0.4149746596813202),
         (-1.0000000150474662e+30, -0.8866188526153564, 0.005409448407590389, 0.5089237093925476, 3.1592581272125244, 0.6291816830635071),
         (-0.5967709422111511, 0.5279380083084106, 1.0000000150474662e+30, 360.0, 2.010882616043091, -0.8510827422142029),
         (-0.8242605924606323, 0.34832876920700073, 0.895243227481842, -2.056257724761963, 1.580925703048706, 0.42089763283729553),
         (0.7087642550468445, 0.2895757257938385, -0.6899518966674805, -0.09705236554145813, 2.2009589672088623, 0.0),
         (0.809302568435669, -0.467088907957077, 0.7051159143447876, 3.0222854614257812, -2.380211353302002, 0.9157808423042297),
         (-0.8258883953094482, 0.7905725240707397, 0.16452465951442719, -0.8904184103012085, -2.445004940032959, 2.401196002960205),
         (-0.19392336905002594, 0.10682975500822067, 0.7816780805587769, -2.6532580852508545, 0.856707751750946, -0.7455148696899414),
         (-0.9212556481361389, 0.9470036029815674, 0.4442795515060425, 0.39398717880249023, 3.02315354347229, 1.388624668121338)]
    ids = [1, 1, 2, 0, 1, 2, 1, 1, 1, 1, 2, 0, 0, 0, 2]
    tfs = [po.Transform(po.vec3(*p[:3]), po.vec3(*p[3:])) for p in P]
    cam = po.Camera(po.vec3(1.0005271035279194e-42, -3.171860456466675, 1.9991309642791748),
                    po.vec3(-0.8964948654174805, 0.5784764289855957, 1.0005271035279194e-42), po.vec3(0, 1, 0), 68.18736359411977)
    light = po.Light(po.vec3(-0.0, -0.0, 1.6980289220809937), po.vec3(0.04442866891622543, -0.10858384519815445, 0.8541004061698914))
    return ids, tfs, cam, light


def test_entries_at_1e30_do_not_escape_the_hierarchy(po, meshes, em_default):
    ids, tfs, cam, light = far_entries_scene(po)
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, cam, light, 200, 40)
    check(em_default.render_instanced(cam, light, ids, tfs, 200, 40), fr, inst, tri)
    assert (fr.shadow == 1).sum() > 2000


def test_no_instances_is_background(po, em_default):
    got = em_default.render_instanced(po.default_camera(), po.default_light(), [], [], 16, 9)
    assert (got["rgb"] == 50).all() and (got["hit_inst"] == -1).all() and (got["shadow"] == 255).all()


# ------------------------------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def gscene(meshes):
    from raytracer_amd import capi
    c = capi.Context(0)
    for m in meshes:
        c.upload_mesh(m.tris, m.nodes, m.rgba)
    yield c
    c.close()


def gpu_render(gscene, po, ids, tfs, w, h, flags=0, block=64):
    import torch
    from raytracer_amd import capi
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    l = po.default_light()
    light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
    inst = capi.Context.make_instances(ids, [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs])
    dev = "cuda:0"
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    hi = torch.empty((h, w), dtype=torch.int32, device=dev)
    ht = torch.empty((h, w), dtype=torch.int32, device=dev)
    t = torch.empty((h, w), dtype=torch.float32, device=dev)
    sh = torch.empty((h, w), dtype=torch.uint8, device=dev)
    gscene.set_option("block", block)
    gscene.render_instanced(cam, light, inst, w, h, rgb, hit_instance=hi, hit_tri=ht, hit_t=t, shadow=sh, flags=flags)
    gscene.sync()
    gscene.set_option("block", 64)
    return {"rgb": rgb.cpu().numpy(), "hit_inst": hi.cpu().numpy(), "hit_tri": ht.cpu().numpy(), "hit_t": t.cpu().numpy(),
            "shadow": sh.cpu().numpy()}


@pytest.mark.gpu
@pytest.mark.parametrize("n_side,size,block", [(1, (240, 135), 64), (5, (333, 211), 256), (10, (160, 90), 64)])
def test_gpu_grid_equals_reference_loop(gscene, po, meshes, n_side, size, block):
    from raytracer_amd import capi
    ids, tfs = grid_scene(po, n_side, seed=n_side)
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), *size)
    got = gpu_render(gscene, po, ids, tfs, *size, flags=capi.RT_FLAG_COUNT_WORK, block=block)
    check(got, fr, inst, tri)
    c = gscene.last_counters()
    assert c["primary_rays"] == size[0] * size[1] and c["hit_pixels"] == fr.counters["hit_pixels"]
    assert c["shadowed_pixels"] == fr.counters["shadowed_pixels"]


@pytest.mark.gpu
def test_gpu_three_instances_are_the_reference_scene(gscene, po, golden_hashes):
    import hashlib
    for f in (0, 45):
        got = gpu_render(gscene, po, [0, 1, 2], po.animation_transforms(f), 1920, 1080)
        want = golden_hashes["frames"][f"1920x1080_f{f}"]
        assert hashlib.sha256(got["rgb"].tobytes()).hexdigest() == want["rgb"]
        assert hashlib.sha256(got["hit_t"].tobytes()).hexdigest() == want["hit_t"]
        assert hashlib.sha256(got["shadow"].tobytes()).hexdigest() == want["shadow"]
        hid = np.where(got["hit_inst"] >= 0, (got["hit_inst"] << 24) | got["hit_tri"], -1).astype(np.int32)
        assert hashlib.sha256(hid.tobytes()).hexdigest() == want["hit_id"]


@pytest.mark.gpu
def test_gpu_large_instance_list_against_cpu_pipeline(gscene, po, meshes, em_default):
    """3 138 instances (about 9.9 million placed triangles): too slow for the reference's per-mesh loop at any useful
    size, so a tile is checked against the oracle and the whole frame against the host build of the same pixel code."""
    ids, tfs = grid_scene(po, 56, seed=3, spacing=1.05)
    assert len(ids) == 56 * 56 + 2
    w, h = 320, 180
    got = gpu_render(gscene, po, ids, tfs, w, h)
    ref = em_default.render_instanced(po.default_camera(), po.default_light(), ids, tfs, w, h)
    for k in ("hit_inst", "hit_tri", "shadow", "rgb"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.array_equal(bits(got["hit_t"]), bits(ref["hit_t"]))
    assert len(np.unique(got["hit_inst"])) > 200
    # and a small frame of the same scene straight against the reference's loop over all 3 138 meshes
    w, h = 96, 54
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), w, h)
    got = gpu_render(gscene, po, ids, tfs, w, h)
    check(got, fr, inst, tri)                             # full list indices (up to 3 137), not their low bits
    assert inst.max() > 255


@pytest.mark.gpu
def test_gpu_fast_build_of_the_instanced_scene_stays_inside_the_stated_tolerance(gscene, po):
    """Context option "fast" on the 3 138-entry list at 1920x1080: fused arithmetic and the tree over the triangles themselves
    (leaves of neighbouring triangles in tight boxes instead of the reference's leaves).  Against the exact build: the same
    (instance, triangle) on at least 99.99 % of camera rays, 8-bit colours within 1 on at least 99.9 % of pixels; and the
    exact build, which stays the default, renders the same bytes afterwards as before."""
    ids, tfs = grid_scene(po, 56, seed=3, spacing=1.05)
    w, h = 1920, 1080
    exact = gpu_render(gscene, po, ids, tfs, w, h)
    gscene.set_option("fast", 1)
    try:
        fast = gpu_render(gscene, po, ids, tfs, w, h)
    finally:
        gscene.set_option("fast", 0)
    same = (fast["hit_inst"] == exact["hit_inst"]) & (fast["hit_tri"] == exact["hit_tri"])
    assert float(same.mean()) >= 0.9999, f"hits differ on {1.0 - float(same.mean()):.6f} of camera rays"
    within = (np.abs(fast["rgb"].astype(np.int16) - exact["rgb"].astype(np.int16)).max(axis=2) <= 1).mean()
    assert float(within) >= 0.999, f"colours within 1 on {float(within):.6f} of pixels"
    assert not np.array_equal(fast["rgb"], exact["rgb"]), "the fast build rendered the exact bytes: is the option reaching the kernels?"
    again = gpu_render(gscene, po, ids, tfs, w, h)
    for k in ("rgb", "hit_inst", "hit_tri", "shadow"):
        assert np.array_equal(again[k], exact[k]), k


@pytest.mark.gpu
def test_gpu_instanced_argument_errors(gscene, po):
    import torch
    from raytracer_amd import capi
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    light = capi.Light(capi.Vec3(0, 1, 0), capi.Vec3(1, 1, 1))
    rgb = torch.empty((8, 8, 3), dtype=torch.uint8, device="cuda:0")
    bad = capi.Context.make_instances([7], [capi.Transform()])
    with pytest.raises(capi.RtError):
        gscene.render_instanced(cam, light, bad, 8, 8, rgb)
    ok = capi.Context.make_instances([0], [capi.Transform()])
    with pytest.raises(capi.RtError):
        gscene.render_instanced(cam, light, ok, 8, 8, rgb, flags=capi.RT_FLAG_REFERENCE_ORDER)
    with pytest.raises(capi.RtError):
        gscene.render_instanced(cam, light, ok, 0, 8, rgb)
    # an empty list renders the background
    none = capi.Context.make_instances([], [])
    gscene.render_instanced(cam, light, none, 8, 8, rgb)
    gscene.sync()
    assert (rgb.cpu().numpy() == 50).all()


@pytest.mark.gpu
def test_gpu_reference_api_scene_with_repeated_meshes(po, meshes):
    """The reference's own calls: one create_mesh per geometry, add_mesh_to_scene many times with different
    placements (scene.c:31-35), render_scene.  More list entries than RT_MAX_MESHES, geometries uploaded once."""
    import ctypes as C
    import os
    from raytracer_amd import host_api as H
    L = H.lib()
    scene = L.create_scene(200, 120, 1000, 4, 1.0)
    L.set_scene_camera(C.byref(scene), H.Vec3(-3, 3, -3), H.Vec3(0, 0, 0), H.Vec3(0, 1, 0), 60.0)
    L.set_scene_light(C.byref(scene), H.Vec3(1, 1, -1), H.Vec3(1.4, 1.4, 1.4))
    loaded = [L.create_mesh(os.path.join(H.ASSETS, n + ".obj").encode(), os.path.join(H.ASSETS, n + ".webp").encode())
              for n in ("drone", "treasure", "ground")]
    ids, tfs = grid_scene(po, 4, seed=9)
    for i, t in zip(ids, tfs):
        m = loaded[i]
        L.set_mesh_position(C.byref(m), H.Vec3(*t.position.tup()))
        L.set_mesh_rotation(C.byref(m), H.Vec3(*t.rotation.tup()))
        L.add_mesh_to_scene(C.byref(scene), m)
    assert scene.mesh_count == 18
    L.render_scene(C.byref(scene))
    assert L.rt_host_last_status(C.byref(scene)) == 0
    want = po.render_frame([meshes[i] for i in ids], tfs, po.default_camera(), po.default_light(), 200, 120, debug=False)
    assert np.array_equal(H.frame_array(scene, 0), want.rgb)
    # moving one entry and rendering the next frame re-uses the uploaded geometries
    L.next_frame(C.byref(scene))
    L.set_mesh_position(C.byref(scene.meshes[5]), H.Vec3(0.3, 1.2, -0.4))
    tfs[5] = po.Transform(po.vec3(0.3, 1.2, -0.4), tfs[5].rotation)
    L.render_scene(C.byref(scene))
    want = po.render_frame([meshes[i] for i in ids], tfs, po.default_camera(), po.default_light(), 200, 120, debug=False)
    assert np.array_equal(H.frame_array(scene, 1), want.rgb)
    for m in loaded:
        L.destroy_mesh(C.byref(m))
    L.destroy_scene(C.byref(scene))


@pytest.mark.gpu
def test_gpu_successive_calls_with_partly_changed_lists(gscene, po, meshes):
    """Placement constants and the hierarchy over the instances are kept between calls: a sequence of
    calls on one context (some entries moved, a mesh swapped, the list shortened and lengthened, a different
    light) gives what the reference loop gives for each list."""
    import torch
    from raytracer_amd import capi
    ids, tfs = grid_scene(po, 6, seed=11)
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    w, h = 160, 90
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    hi = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
    rng = np.random.default_rng(5)
    lights = [po.default_light(), po.lib().orc_make_light(po.vec3(-0.3, 1, 0.4), po.vec3(1.0, 0.9, 0.8))]
    steps = []
    cur_ids, cur_tfs = list(ids), list(tfs)
    steps.append((list(cur_ids), list(cur_tfs), 0))
    for k in (3, 7, 20):                                                         # a few entries move
        cur_tfs[k] = po.Transform(po.vec3(*rng.uniform(-2, 2, 3) * [1, 0.2, 1] + [0, 0.8, 0]), po.vec3(0, float(rng.uniform(0, 6)), 0))
    steps.append((list(cur_ids), list(cur_tfs), 0))
    cur_ids[5] = 0 if cur_ids[5] != 0 else 1                                     # same transform, another mesh
    steps.append((list(cur_ids), list(cur_tfs), 0))
    steps.append((cur_ids[:10], cur_tfs[:10], 0))                                # shorter, then the long list again
    steps.append((list(cur_ids), list(cur_tfs), 0))
    steps.append((list(cur_ids), list(cur_tfs), 1))                              # another light: every constant changes
    steps.append((list(ids), list(tfs), 0))
    fly = list(tfs)
    for f in (20, 30, 40):                                                       # one mover per frame: the kept hierarchy is patched
        fly[0] = po.animation_transforms(f)[0]
        steps.append((list(ids), list(fly), 0))
    for s_ids, s_tfs, li in steps:
        l = lights[li]
        light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
        inst = capi.Context.make_instances(s_ids, [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in s_tfs])
        gscene.render_instanced(cam, light, inst, w, h, rgb, hit_instance=hi)
        gscene.sync()
        fr, inst_want, _ = oracle_planes(po, meshes, s_ids, s_tfs, po.default_camera(), l, w, h)
        assert np.array_equal(hi.cpu().numpy(), inst_want)
        assert np.array_equal(rgb.cpu().numpy(), fr.rgb)


@pytest.mark.gpu
def test_gpu_frame_without_a_shadow_plane_has_the_same_colours(gscene, po, meshes):
    """A call that asks for no shadow plane sends no shadow ray where it cannot change the colour (surfaces whose own term
    max(0.2, n.L) is 0.2 already): colour, instance, triangle and distance planes stay the reference loop's."""
    import torch
    from raytracer_amd import capi
    ids, tfs = grid_scene(po, 6, seed=3)
    w, h = 321, 187
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), w, h)
    cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
    l = po.default_light()
    light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
    placed = capi.Context.make_instances(ids, [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs])
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    hi = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
    ht = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
    gscene.render_instanced(cam, light, placed, w, h, rgb, hit_instance=hi, hit_tri=ht)
    gscene.sync()
    assert np.array_equal(rgb.cpu().numpy(), fr.rgb)
    assert np.array_equal(hi.cpu().numpy(), inst) and np.array_equal(ht.cpu().numpy()[inst >= 0], tri[inst >= 0])


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", [1, 4])
def test_gpu_wave_scheduled_kernel_equals_reference_loop(gscene, po, meshes, tiles):
    """Option "instanced_wave": the per-wavefront scheduler (every lane a state machine, lanes refilled from a strip of
    tiles) gives the planes of the reference loop -- grids of instances at ragged sizes, coincident instances (the earlier
    list entry wins an equal distance), and the reference scene's full-size hashes."""
    gscene.set_option("instanced_wave", tiles)
    try:
        for n_side, size in ((1, (240, 135)), (5, (333, 211)), (10, (161, 93))):
            ids, tfs = grid_scene(po, n_side, seed=n_side)
            fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), *size)
            check(gpu_render(gscene, po, ids, tfs, *size), fr, inst, tri)
            # the treasures alone: drone and ground hold large triangles, and a list that places one of them goes to the
            # default kernel whatever the option says -- without them the scheduler itself renders
            fr, inst, tri = oracle_planes(po, meshes, ids[2:], tfs[2:], po.default_camera(), po.default_light(), *size)
            check(gpu_render(gscene, po, ids[2:], tfs[2:], *size), fr, inst, tri)
        t = po.animation_transforms(7)
        ids, tfs = [1, 0, 1, 2, 1], [t[1], t[0], t[1], t[2], t[1]]                    # the treasure three times in one place
        fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), 200, 120)
        got = gpu_render(gscene, po, ids, tfs, 200, 120)
        check(got, fr, inst, tri)
        assert 0 in inst and 2 not in inst and 4 not in inst
    finally:
        gscene.set_option("instanced_wave", 0)


@pytest.mark.gpu
def test_gpu_entries_with_nan_and_infinite_placements(gscene, po, meshes):
    """nonfinite_lists on the GPU, one call after the other on one context (kept hierarchy, kept placement constants)."""
    for name, ids, tfs in nonfinite_lists(po):
        fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), 96, 54)
        check(gpu_render(gscene, po, ids, tfs, 96, 54), fr, inst, tri)


@pytest.mark.gpu
def test_gpu_entries_at_1e30_do_not_escape_the_hierarchy(gscene, po, meshes):
    import torch
    from raytracer_amd import capi
    ids, tfs, cam, l = far_entries_scene(po)
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, cam, l, 200, 40)
    ccam = capi.Camera(capi.Vec3(*cam.position.tup()), capi.Vec3(*cam.look_at.tup()), capi.Vec3(*cam.up.tup()), cam.fov)
    light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
    lst = capi.Context.make_instances(ids, [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs])
    rgb = torch.empty((40, 200, 3), dtype=torch.uint8, device="cuda:0")
    hi = torch.empty((40, 200), dtype=torch.int32, device="cuda:0")
    sh = torch.empty((40, 200), dtype=torch.uint8, device="cuda:0")
    gscene.render_instanced(ccam, light, lst, 200, 40, rgb, hit_instance=hi, shadow=sh)
    gscene.sync()
    assert np.array_equal(hi.cpu().numpy(), inst) and np.array_equal(sh.cpu().numpy(), fr.shadow) and np.array_equal(rgb.cpu().numpy(), fr.rgb)
