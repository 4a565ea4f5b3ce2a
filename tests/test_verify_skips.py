"""RT_FLAG_VERIFY_SKIPS: the kernels check their own shortcuts.

Everything the product skips by geometry -- meshes left out per tile by screen and shadow rectangles, walks entered through a
flat mesh's grid, subtrees dropped by the distance cull, the second tree, instances the hierarchy does not propose -- rests on
margins for float rounding.  With the flag the pixel code does, beside every shortcut, what the reference would have done
(scene.c:72-81, 124-139; bvh.c:80-129; aabb.c:33-56) on the very same ray and counts contradictions.  These tests render the
golden frames, the scenes aimed at the margins, the degenerate views, the non-finite lists and the 3 138-entry list with the
flag and require zero contradictions and the reference's planes; and they damage a rectangle, a grid and a second tree on
purpose to see the counters move.  CPU: the host build of the pixel code.  GPU: through the C ABI."""
import numpy as np
import pytest

import emul
from conftest import bits
from test_host_emul import checker, _flat_mesh, tri_rows
from test_instanced import grid_scene, oracle_planes, nonfinite_lists, far_entries_scene, check as check_instanced

KINDS = ("vis_disagreements", "shadow_vis_disagreements", "entry_disagreements", "cull_disagreements", "instance_disagreements")


def assert_clean(counters, what, min_checks=1):
    bad = {k: counters[k] for k in KINDS if counters[k]}
    assert not bad, f"{what}: the reference's way contradicts the shortcuts: {bad}"
    assert counters["skip_checks"] >= min_checks, f"{what}: only {counters['skip_checks']} decisions were examined"


def assert_planes(got, want, what):
    assert np.array_equal(got["hit_id"], want.hit_id), f"{what}: hit ids"
    assert np.array_equal(bits(got["hit_t"]), bits(want.hit_t)), f"{what}: t"
    assert np.array_equal(got["shadow"], want.shadow) and np.array_equal(got["rgb"], want.rgb), f"{what}: shading"


@pytest.fixture(scope="module")
def em_scene(meshes):
    s = emul.Scene(regroup=True)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    yield s
    s.close()


# ------------------------------------------------------------------------------------------ CPU: host build of the pixel code
@pytest.mark.parametrize("frame", [0, 45, 95])
def test_golden_frames_check_themselves(po, em_scene, golden_frames, frame):
    got = em_scene.render(po.default_camera(), po.default_light(), po.animation_transforms(frame), 240, 135, verify=True)
    for k in ("hit_id", "shadow", "rgb"):
        assert np.array_equal(got[k], golden_frames[f"f{frame}_{k}"]), k
    assert np.array_equal(bits(got["hit_t"]), bits(golden_frames[f"f{frame}_hit_t"]))
    # every query is walked three ways (two comparisons), and the tiles the rectangles leave out get their root tests
    assert_clean(got["counters"], f"frame {frame}", min_checks=240 * 135 * 6)


def test_scenes_aimed_at_the_margins_check_themselves(po):
    import adversarial
    tex = checker()
    n = 0
    for name, soups, tfs, cam, w, h in adversarial.scenes(po, seed=5):
        hm = adversarial.build(po, soups, tex)
        s = emul.Scene(regroup=True)
        for m in hm:
            assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
        got = s.render(cam, po.default_light(), tfs, w, h, verify=True)
        assert_planes(got, po.render_frame(hm, tfs, cam, po.default_light(), w, h), name)
        assert_clean(got["counters"], name)
        s.close()
        n += 1
    assert n >= 20


def test_degenerate_views_check_themselves(po, meshes, em_scene):
    import adversarial
    n = 0
    for name, tfs, cam, light in adversarial.degenerate_views(po):
        got = em_scene.render(cam, light, tfs, 96, 54, verify=True)
        assert_planes(got, po.render_frame(meshes, tfs, cam, light, 96, 54), name)
        assert_clean(got["counters"], name)
        n += 1
    assert n >= 20


def flat_scenes(po, trials=9):
    """Sheets of zero thickness along x, y, z under random placements, seen steeply and at grazing angles (as
    test_flat_meshes_entered_through_their_grid_give_the_reference_frame builds them)."""
    rng = np.random.default_rng(23)
    tex = checker()
    for trial in range(trials):
        axis = trial % 3
        extent = float(10 ** rng.uniform(-0.3, 1.0))
        offset = (rng.normal(size=3) * (1.0e3 if trial % 6 >= 4 else 1.0)).astype(np.float32).astype(np.float64)
        sheet = _flat_mesh(rng, axis, int(rng.choice([6, 12, 20])), extent, offset)
        blob = tri_rows((offset + rng.normal(size=(40, 1, 3)) * extent * 0.3 + rng.normal(size=(40, 3, 3)) * extent * 0.05).astype(np.float32))
        tfs = [po.Transform(po.vec3(*rng.normal(size=3)), po.vec3(*rng.uniform(-1, 1, size=3))), po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0))]
        nrm = np.zeros(3); nrm[axis] = 1.0
        side = rng.normal(size=3); side[axis] = 0; side /= np.linalg.norm(side)
        dist = extent * float(rng.uniform(0.5, 4.0))
        eye_local = offset + side * dist * 0.9 + nrm * dist * [0.8, 0.05, 1e-3][trial % 3]
        M = np.linalg.inv(np.array(po.mesh_matrices(tfs[0])[0], np.float64))
        eye, look = (M @ np.append(eye_local, 1.0))[:3], (M @ np.append(offset, 1.0))[:3]
        cam = po.Camera(po.vec3(*eye), po.vec3(*look), po.vec3(*(M[:3, :3] @ nrm)), float(rng.choice([20.0, 60.0])))
        ld = rng.normal(size=3); ld /= np.linalg.norm(ld)
        light = po.Light(po.vec3(*np.array(ld, np.float32)), po.vec3(1.2, 1.1, 1.0))
        hm = []
        for soup in (sheet, blob):
            t, nodes = po.build_bvh(soup)
            hm.append(po.HostMesh(t, nodes, tex))
        yield f"sheet {trial} along axis {axis}", hm, tfs, cam, light


def test_flat_meshes_check_their_grid_entries(po):
    for name, hm, tfs, cam, light in flat_scenes(po):
        s = emul.Scene(regroup=True)
        for m in hm:
            assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
        got = s.render(cam, light, tfs, 128, 72, verify=True)
        assert_planes(got, po.render_frame(hm, tfs, cam, light, 128, 72), name)
        assert_clean(got["counters"], name)
        s.close()


def test_sampled_frame_checks_itself(po, em_scene):
    l = po.default_light()
    lights = [l, po.Light(po.vec3(-0.5, 0.6, 0.62449980), po.vec3(0.3, 0.5, 0.9))]
    got = em_scene.render_sampled(po.default_camera(), lights, po.animation_transforms(45), 4, 11, 120, 68, verify=True)
    ref = em_scene.render_sampled(po.default_camera(), lights, po.animation_transforms(45), 4, 11, 120, 68)
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.array_equal(got["hit_id"], ref["hit_id"])
    assert_clean(got["counters"], "sampled frame", min_checks=120 * 68 * 4 * 6)


def test_instance_lists_check_their_hierarchy(po, meshes, em_scene):
    for n_side, size in ((1, (96, 54)), (5, (120, 68)), (10, (96, 54))):
        ids, tfs = grid_scene(po, n_side, seed=n_side)
        fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), *size)
        got = em_scene.render_instanced(po.default_camera(), po.default_light(), ids, tfs, *size, verify=True)
        check_instanced(got, fr, inst, tri)
        assert_clean(got["counters"], f"{len(ids)} entries", min_checks=size[0] * size[1])
    for name, ids, tfs in nonfinite_lists(po):
        got = em_scene.render_instanced(po.default_camera(), po.default_light(), ids, tfs, 64, 36, verify=True)
        assert_clean(got["counters"], name)
    ids, tfs, cam, light = far_entries_scene(po)
    got = em_scene.render_instanced(cam, light, ids, tfs, 200, 40, verify=True)
    assert_clean(got["counters"], "entries at 1e30")


def test_thousands_of_entries_check_their_hierarchy(po, em_scene):
    ids, tfs = grid_scene(po, 56, seed=3, spacing=1.05)
    got = em_scene.render_instanced(po.default_camera(), po.default_light(), ids, tfs, 48, 27, verify=True)
    assert_clean(got["counters"], "3 138 entries", min_checks=48 * 27)
    assert got["hit_inst"].max() > 255


def test_damaged_shortcuts_are_noticed(po, meshes):
    """The counters move when a shortcut is wrong: rectangles two tiles too small, a grid whose every cell names one leaf,
    a second tree that has lost half of itself."""
    cam, light, tfs = po.default_camera(), po.default_light(), po.animation_transforms(45)
    s = emul.Scene(regroup=True)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    clean = s.render(cam, light, tfs, 240, 135, verify=True)["counters"]
    assert_clean(clean, "undamaged")
    s.set_vis_shrink(2)
    c = s.render(cam, light, tfs, 240, 135, verify=True)["counters"]
    assert c["vis_disagreements"] > 100 and c["shadow_vis_disagreements"] > 100, c
    assert c["entry_disagreements"] == 0 and c["cull_disagreements"] == 0
    s.set_vis_shrink(0)
    assert s.damage_grid(2) == 0                      # the ground is the flat mesh
    c = s.render(cam, light, tfs, 240, 135, verify=True)["counters"]
    assert c["entry_disagreements"] > 1000 and c["cull_disagreements"] == 0 and c["vis_disagreements"] == 0, c
    s.close()
    s = emul.Scene(regroup=True)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    assert s.damage_second_tree(1) == 0               # the treasure
    c = s.render(cam, light, tfs, 240, 135, verify=True)["counters"]
    assert c["cull_disagreements"] > 100 and c["entry_disagreements"] == 0, c
    s.close()
    s = emul.Scene(regroup=True)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    assert s.damage_grid_masks(2) == 0                # the ground's cells name the wrong triangles
    c = s.render(cam, light, tfs, 240, 135, verify=True)["counters"]
    assert c["entry_disagreements"] > 1000 and c["cull_disagreements"] == 0 and c["vis_disagreements"] == 0, c
    s.close()
    s = emul.Scene(regroup=True)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    assert s.grid_axis(0) == 5 and s.damage_shear(0) == 0      # the drone's tree in the sheared frame has lost half of itself
    c = s.render(cam, light, tfs, 240, 135, verify=True)["counters"]
    assert c["entry_disagreements"] > 100 and c["cull_disagreements"] == 0 and c["vis_disagreements"] == 0, c
    s.close()


def test_meshes_of_diagonal_beams_check_their_sheared_tree(po):
    """tests/adversarial.py::diagonal_frames: the packer gives those meshes a tree in the sheared frame of their plane; from
    every view -- rays the tree may take, rays that test some families of triangles apart, rays that have to take the other
    trees -- the frame is the oracle's and the walk through that tree ends like the walk from the root."""
    import adversarial
    tex = checker()
    n = 0
    for name, soups, tfs, cam, light, w, h, kept in adversarial.diagonal_frames(po):
        hm = adversarial.build(po, soups, tex)
        s = emul.Scene(regroup=True)
        for m in hm:
            assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
        assert s.grid_axis(0) == 4 + kept and s.grid_axis(1) < 4, name
        got = s.render(cam, light, tfs, w, h, verify=True)
        assert_planes(got, po.render_frame(hm, tfs, cam, light, w, h), name)
        assert_clean(got["counters"], name)
        s.close()
        n += 1
    assert n >= 40


# ------------------------------------------------------------------------------------------ GPU: through the C ABI
@pytest.fixture(scope="module")
def capi():
    from raytracer_amd import capi as mod
    mod.lib()
    return mod


def conv(capi, tfs):
    return [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in tfs]


def ccam(capi, cam):
    return capi.Camera(capi.Vec3(*cam.position.tup()), capi.Vec3(*cam.look_at.tup()), capi.Vec3(*cam.up.tup()), cam.fov)


def clight(capi, l):
    return capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))


def gpu_checked(capi, c, cam, light, tfs, w, h):
    rgb, hid, ht, sh = c.render_host(ccam(capi, cam), clight(capi, light), conv(capi, tfs), w, h, flags=capi.RT_FLAG_VERIFY_SKIPS)
    return {"rgb": rgb, "hit_id": hid, "hit_t": ht, "shadow": sh, "counters": c.last_counters()}


@pytest.fixture(scope="module")
def gctx(capi, meshes):
    c = capi.Context(0)
    for m in meshes:
        c.upload_mesh(m.tris, m.nodes, m.rgba)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("frame", [0, 45, 95])
def test_gpu_golden_frames_check_themselves(capi, gctx, po, golden_frames, frame):
    got = gpu_checked(capi, gctx, po.default_camera(), po.default_light(), po.animation_transforms(frame), 240, 135)
    for k in ("hit_id", "shadow", "rgb"):
        assert np.array_equal(got[k], golden_frames[f"f{frame}_{k}"]), k
    assert np.array_equal(bits(got["hit_t"]), bits(golden_frames[f"f{frame}_hit_t"]))
    assert_clean(got["counters"], f"frame {frame}", min_checks=240 * 135 * 6)
    assert got["counters"]["primary_rays"] == 240 * 135


@pytest.mark.gpu
def test_gpu_full_size_frames_check_themselves(capi, gctx, po, golden_hashes):
    """1920x1080, frames 0 and 45: the planes hash like the compiled reference's and no shortcut is contradicted on any of the
    2 073 600 pixels -- the size the bench runs at, where the rectangles and the grid see other magnitudes than at 240x135."""
    import hashlib
    for f in (0, 45):
        got = gpu_checked(capi, gctx, po.default_camera(), po.default_light(), po.animation_transforms(f), 1920, 1080)
        want = golden_hashes["frames"][f"1920x1080_f{f}"]
        for k in ("rgb", "hit_id", "hit_t", "shadow"):
            assert hashlib.sha256(got[k].tobytes()).hexdigest() == want[k], (f, k)
        assert_clean(got["counters"], f"1920x1080 frame {f}", min_checks=1920 * 1080 * 6)


@pytest.mark.gpu
def test_gpu_scenes_aimed_at_the_margins_check_themselves(capi, po):
    import adversarial
    tex = checker()
    n = 0
    for name, soups, tfs, cam, w, h in adversarial.scenes(po, seed=5):
        hm = adversarial.build(po, soups, tex)
        with capi.Context(0) as c:
            for m in hm:
                c.upload_mesh(m.tris, m.nodes, m.rgba)
            got = gpu_checked(capi, c, cam, po.default_light(), tfs, w, h)
        assert_planes(got, po.render_frame(hm, tfs, cam, po.default_light(), w, h), name)
        assert_clean(got["counters"], name)
        n += 1
    assert n >= 20


def test_per_frame_masks_of_the_sheared_tree_never_hide_a_close_family(po, meshes):
    """Which families of the drone's triangles a ray runs too close to is decided per ray (shear_gate); per frame the host
    answers it for the shadow rays' direction and rules families out for all camera rays through the root box
    (shear_frame_masks).  Over the animation and 300 random placements, cameras and lights: whatever the gate flags when it
    looks at every family, it flags with the masks too -- and the masks do spare it work."""
    s = emul.Scene(regroup=True)
    for m in meshes:
        assert s.add(m.tris, m.nodes, m.rgba) == 0, s.error()
    assert s.grid_axis(0) == 5
    looked = spared = 0
    for frame in (0, 24, 48, 72, 95):
        bad, n, k = s.shear_masks_hold(0, po.default_camera(), po.default_light(), po.animation_transforms(frame)[0], 480, 270)
        assert bad == 0, frame
        looked += n; spared += k
    rng = np.random.default_rng(3)
    for _ in range(300):
        tf = po.Transform(po.vec3(*rng.uniform(-2, 2, 3)), po.vec3(*rng.uniform(-3.2, 3.2, 3)))
        cam = po.Camera(po.vec3(*rng.uniform(-5, 5, 3)), po.vec3(*(np.array(tf.position.tup()) + rng.normal(0, 0.5, 3))), po.vec3(0, 1, 0),
                        float(rng.uniform(20, 100)))
        light = po.lib().orc_make_light(po.vec3(*rng.uniform(-1, 1, 3)), po.vec3(1, 1, 1))
        bad, n, k = s.shear_masks_hold(0, cam, light, tf, 96, 54)
        assert bad == 0
        looked += n; spared += k
    assert looked > 100000 and spared > 10 * looked
    s.close()


@pytest.mark.gpu
def test_gpu_meshes_of_diagonal_beams_check_their_sheared_tree(capi, po):
    """The same scenes through the C ABI: the self-checking frame, the plain frame with all planes and the batched frames."""
    import torch
    import adversarial
    tex = checker()
    n = 0
    for name, soups, tfs, cam, light, w, h, kept in adversarial.diagonal_frames(po):
        hm = adversarial.build(po, soups, tex)
        want = po.render_frame(hm, tfs, cam, light, w, h)
        cl = clight(capi, light)
        with capi.Context(0) as c:
            for m in hm:
                c.upload_mesh(m.tris, m.nodes, m.rgba)
            got = gpu_checked(capi, c, cam, light, tfs, w, h)
            assert_planes(got, want, name)
            assert_clean(got["counters"], name)
            rgb, hid, ht, sh = c.render_host(ccam(capi, cam), cl, conv(capi, tfs), w, h)
            assert np.array_equal(rgb, want.rgb) and np.array_equal(hid, want.hit_id) and np.array_equal(sh, want.shadow), name
            assert np.array_equal(ht.view(np.uint32), want.hit_t.view(np.uint32)), name
            out = torch.zeros((2, h, w, 3), dtype=torch.uint8, device="cuda:0")
            c.render_frames(ccam(capi, cam), cl, [conv(capi, tfs)] * 2, w, h, out)
            c.sync()
            assert np.array_equal(out[0].cpu().numpy(), want.rgb) and np.array_equal(out[1].cpu().numpy(), want.rgb), name
        n += 1
    assert n >= 40


@pytest.mark.gpu
def test_gpu_degenerate_views_and_flat_meshes_check_themselves(capi, gctx, po, meshes):
    import adversarial
    n = 0
    for name, tfs, cam, light in adversarial.degenerate_views(po):
        got = gpu_checked(capi, gctx, cam, light, tfs, 96, 54)
        assert_planes(got, po.render_frame(meshes, tfs, cam, light, 96, 54), name)
        assert_clean(got["counters"], name)
        n += 1
    assert n >= 20
    for name, hm, tfs, cam, light in flat_scenes(po):
        with capi.Context(0) as c:
            for m in hm:
                c.upload_mesh(m.tris, m.nodes, m.rgba)
            got = gpu_checked(capi, c, cam, light, tfs, 128, 72)
        assert_planes(got, po.render_frame(hm, tfs, cam, light, 128, 72), name)
        assert_clean(got["counters"], name)


@pytest.mark.gpu
def test_gpu_nonfinite_meshes_check_themselves(capi, po):
    import adversarial
    tex = checker()
    ident = [po.Transform(po.vec3(0, 0, 0), po.vec3(0.2, 0.4, 0.1))]
    n = 0
    for name, soup in adversarial.nonfinite_soups(sizes=(7, 60, 500)):
        t, nodes = capi.build_bvh(soup)
        with capi.Context(0) as c:
            c.upload_mesh(t, nodes, tex)
            got = gpu_checked(capi, c, po.default_camera(), po.default_light(), ident, 96, 54)
        assert_planes(got, po.render_frame([po.HostMesh(t, nodes, tex)], ident, po.default_camera(), po.default_light(), 96, 54), name)
        assert_clean(got["counters"], name)
        n += 1
    assert n >= 30


@pytest.mark.gpu
def test_gpu_sampled_frame_checks_itself(capi, gctx, po):
    import torch
    w, h = 240, 135
    l = po.default_light()
    lights = [clight(capi, l), capi.Light(capi.Vec3(-0.5, 0.6, 0.62449980), capi.Vec3(0.3, 0.5, 0.9))]
    a = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    b = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    tfs = conv(capi, po.animation_transforms(45))
    gctx.render_sampled(ccam(capi, po.default_camera()), lights, tfs, w, h, 8, 5, a, flags=capi.RT_FLAG_VERIFY_SKIPS)
    cnt = gctx.last_counters()
    gctx.render_sampled(ccam(capi, po.default_camera()), lights, tfs, w, h, 8, 5, b)
    gctx.sync()
    assert torch.equal(a, b)
    assert_clean(cnt, "sampled frame", min_checks=w * h * 8 * 6)
    assert cnt["primary_rays"] == w * h * 8


def gpu_instanced_checked(capi, c, po, ids, tfs, cam, light, w, h):
    import torch
    inst = capi.Context.make_instances(ids, conv(capi, tfs))
    dev = "cuda:0"
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    hi = torch.zeros((h, w), dtype=torch.int32, device=dev)
    ht = torch.zeros((h, w), dtype=torch.int32, device=dev)
    t = torch.zeros((h, w), dtype=torch.float32, device=dev)
    sh = torch.zeros((h, w), dtype=torch.uint8, device=dev)
    c.render_instanced(ccam(capi, cam), clight(capi, light), inst, w, h, rgb, hit_instance=hi, hit_tri=ht, hit_t=t, shadow=sh,
                       flags=capi.RT_FLAG_VERIFY_SKIPS)
    cnt = c.last_counters()
    return {"rgb": rgb.cpu().numpy(), "hit_inst": hi.cpu().numpy(), "hit_tri": ht.cpu().numpy(), "hit_t": t.cpu().numpy(),
            "shadow": sh.cpu().numpy(), "counters": cnt}


@pytest.mark.gpu
def test_gpu_instance_lists_check_their_hierarchy(capi, gctx, po, meshes):
    cam, light = po.default_camera(), po.default_light()
    for n_side, size in ((1, (240, 135)), (5, (160, 90)), (10, (120, 68))):
        ids, tfs = grid_scene(po, n_side, seed=n_side)
        fr, inst, tri = oracle_planes(po, meshes, ids, tfs, cam, light, *size)
        got = gpu_instanced_checked(capi, gctx, po, ids, tfs, cam, light, *size)
        check_instanced(got, fr, inst, tri)
        assert_clean(got["counters"], f"{len(ids)} entries", min_checks=size[0] * size[1])
    for name, ids, tfs in nonfinite_lists(po):
        fr, inst, tri = oracle_planes(po, meshes, ids, tfs, cam, light, 96, 54)
        got = gpu_instanced_checked(capi, gctx, po, ids, tfs, cam, light, 96, 54)
        check_instanced(got, fr, inst, tri)
        assert_clean(got["counters"], name)
    ids, tfs, fcam, flight = far_entries_scene(po)
    got = gpu_instanced_checked(capi, gctx, po, ids, tfs, fcam, flight, 200, 40)
    assert_clean(got["counters"], "entries at 1e30")


@pytest.mark.gpu
def test_gpu_thousands_of_entries_check_their_hierarchy(capi, gctx, po, meshes):
    """3 138 list entries (9.9 million placed triangles) at 96x54: the frame equals the reference's loop over the whole list
    (run by the oracle) and, ray by ray on the device, the hierarchy's answer equals that loop's."""
    ids, tfs = grid_scene(po, 56, seed=3, spacing=1.05)
    w, h = 96, 54
    fr, inst, tri = oracle_planes(po, meshes, ids, tfs, po.default_camera(), po.default_light(), w, h)
    got = gpu_instanced_checked(capi, gctx, po, ids, tfs, po.default_camera(), po.default_light(), w, h)
    check_instanced(got, fr, inst, tri)
    assert_clean(got["counters"], "3 138 entries", min_checks=w * h)


# What the long self-checking runs found on tame input (tests/golden/cull_findings.json: three scenes of 90 000, one ray each):
# the driver's meshes under random placements, the camera close to the drone.  In the first, the ray of pixel (596, 127) runs
# within 2e-4 rad of the plane of one of the drone's 2.8-unit rotor triangles and misses the triangle (v = -0.07 in double
# precision); the reference's float triangle test (ray.c:42-67), with a = -2.9e-6, lets it pass with t = 1.6667, 8 % off the
# plane's true distance of 1.811 -- in front of the real surface at 1.7067, which the ordered walk had found by then, so that it
# dropped the rotor's box (entered beyond 1.71) unvisited.  The other two are the same triangles under other views.  Since then
# nothing of a mesh with such triangles is skipped by distance (MeshArgs::wide), and these scenes are the regression test.
def cull_finding_scenes(po):
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cull_findings.json")) as f:
        records = json.load(f)
    f32 = lambda v: float(np.float32(float(v)))
    for r in records:
        c = [f32(v) for v in r["camera"]]
        l = [f32(v) for v in r["light"]]
        cam = po.Camera(po.vec3(*c[0:3]), po.vec3(*c[3:6]), po.vec3(*c[6:9]), c[9])
        light = po.Light(po.vec3(*l[0:3]), po.vec3(*l[3:6]))
        tfs = [po.Transform(po.vec3(*[f32(v) for v in p[0:3]]), po.vec3(*[f32(v) for v in p[3:6]])) for p in r["placements"]]
        yield f"scene {r['scene']} of seed {r['seed']}", cam, light, tfs, tuple(r["size"])


def test_which_meshes_are_walked_without_the_distance_cull(po, meshes, em_scene):
    """The drone (rotor discs of 2.8 x 2.8) and the ground (quads of 2 x 2) hold wide triangles, the treasure (|e1||e2| up to 0.12)
    does not; the same drone at a tenth of its size does not either, and a NaN vertex counts as wide."""
    assert [em_scene.wide(i) for i in range(3)] == [True, False, True]
    s = emul.Scene(regroup=True)
    small = meshes[0].tris.copy()
    small[:, :9] *= np.float32(0.1)
    t, n = po.build_bvh(small)
    assert s.add(t, n, meshes[0].rgba) == 0
    broken = meshes[1].tris.copy()
    broken[7, 4] = np.float32("nan")
    t, n = po.build_bvh(broken)
    assert s.add(t, n, meshes[1].rgba) == 0
    assert [s.wide(0), s.wide(1)] == [False, True]
    s.close()


def test_scenes_in_which_the_distance_cull_once_changed_a_pixel(po, meshes, em_scene):
    """CPU build of the pixel code, default walk: every plane equals the reference's again."""
    n = 0
    for name, cam, light, tfs, (w, h) in cull_finding_scenes(po):
        got = em_scene.render(cam, light, tfs, w, h, screen_cull=True)
        assert_planes(got, po.render_frame(meshes, tfs, cam, light, w, h), name)
        n += 1
    assert n == 3


@pytest.mark.gpu
def test_gpu_scenes_in_which_the_distance_cull_once_changed_a_pixel(capi, gctx, po, meshes):
    """Through the C ABI with the self-check on: no disagreement of any kind, planes equal to the reference's."""
    for name, cam, light, tfs, (w, h) in cull_finding_scenes(po):
        got = gpu_checked(capi, gctx, cam, light, tfs, w, h)
        assert_planes(got, po.render_frame(meshes, tfs, cam, light, w, h), name)
        assert_clean(got["counters"], name, min_checks=w * h)
