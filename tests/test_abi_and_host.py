"""CPU-side checks of the product libraries: the C-ABI library loads and exports every symbol its
header declares, the host-only entry points (BVH build, tree flattening, mesh loading) reproduce the
reference's results, and rendering entry points fail loudly -- never fall back -- without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import REPO, bits

NAMES = ("drone", "treasure", "ground")
HEADER = os.path.join(REPO, "include", "rt_mi355.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def gpu_present():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="module")
def capi():
    from raytracer_amd import capi as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def host():
    from raytracer_amd import host_api as m
    m.lib()
    return m


def test_library_exports_every_declared_symbol(capi):
    names = declared_functions()
    assert len(names) >= 19 and "rt_render_frame" in names and "rt_render_frames" in names
    L = capi.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rt_mi355.h but not exported"
    assert sorted(capi.EXPORTS) == names                     # the Python binding covers the whole header
    # and the dynamic symbol table holds them as C symbols (no C++ mangling)
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (rt_[a-z_0-9]+)$", out, flags=re.M))
    assert set(names) <= exported


def test_header_compiles_as_c():
    src = '#include "rt_mi355.h"\nint main(void){ rt_counters c; rt_flat_node n; (void)c; (void)n; return sizeof(rt_triangle) == 96 ? 0 : 1; }\n'
    exe = "/tmp/rt_header_check"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(REPO, "include"), "-x", "c", "-", "-o", exe],
                   input=src, text=True, check=True)
    assert subprocess.run([exe]).returncode == 0


def test_host_library_exports_the_reference_api(host):
    """Names of scene.h:29-36, geometry/mesh.h:22-26, accel/bvh.h:23-28, render/camera.h:15, render/light.h:12."""
    for n in ("create_scene", "add_mesh_to_scene", "set_scene_camera", "set_scene_light", "next_frame", "render_scene",
              "save_scene", "destroy_scene", "create_mesh", "set_mesh_position", "set_mesh_rotation", "destroy_mesh",
              "sample_mesh_texture", "create_bvh_node", "create_bvh", "destroy_bvh_node", "destroy_bvh", "intersect_bvh",
              "create_camera", "create_directional_light"):
        assert hasattr(host.lib(), n), n


def test_status_strings(capi):
    assert capi.status_string(0) == "ok"
    for st in (capi.RT_ERR_INVALID, capi.RT_ERR_DEVICE, capi.RT_ERR_NO_DEVICE, capi.RT_ERR_UNSUPPORTED, capi.RT_ERR_CAPACITY):
        assert capi.status_string(st) not in ("ok", "unknown status")
    assert capi.status_string(-99) == "unknown status"


def test_c_driver_on_the_host_api_stops_without_a_gpu(host, tmp_path):
    """examples/drone_animation.c is plain C11 on raytracer_host.h (built by `make examples`): without a GPU it loads the
    meshes, render_scene reports why it cannot render, and the program stops without writing a file."""
    exe = os.path.join(REPO, "examples", "drone_animation")
    assert os.path.exists(exe), "run `make examples`"
    if gpu_present():
        pytest.skip("a GPU is present")
    out = str(tmp_path / "never.webp")
    r = subprocess.run([exe, host.ASSETS, out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "no usable GPU" in r.stderr and "frame 0 failed" in r.stderr
    assert not os.path.exists(out) and not os.path.exists(out + ".ppm")


def test_host_library_cpu_side_under_sanitizers(host, tmp_path):
    """tests/host_emul/host_check.cpp compiled together with the host library's sources under the address and
    undefined-behaviour sanitizers: the OBJ reader in both modes on the assets, on malformed files and on randomly
    damaged copies of a real file; unreadable textures; the BVH build on tiny, degenerate and non-finite input; the
    animation writer.  No invalid access, no undefined operation, no leak, no data race (second build, thread sanitizer), and the
    program's own checks hold."""
    src = os.path.join(REPO, "raytracer_amd", "host")
    exe = str(tmp_path / "host_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-static-libasan", "-static-libubsan", "-I", os.path.join(REPO, "include"), "-o", exe,
                    os.path.join(REPO, "tests", "host_emul", "host_check.cpp"), os.path.join(src, "raytracer_host.cpp"),
                    os.path.join(src, "raytracer_math.cpp"), os.path.join(src, "webp_anim.cpp"), "-pthread",
                    "-L", os.path.join(REPO, "raytracer_amd", "csrc"), "-lrt_mi355", "-ldl", "-lm",
                    "-Wl,-rpath," + os.path.join(REPO, "raytracer_amd", "csrc")], check=True, timeout=600)
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    r = subprocess.run([exe, host.ASSETS, str(scratch)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == '{"errors": 0}'
    # the same program under the thread sanitizer: the animation writer encodes on worker threads
    exe_t = str(tmp_path / "host_check_tsan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=thread", "-I", os.path.join(REPO, "include"), "-o", exe_t,
                    os.path.join(REPO, "tests", "host_emul", "host_check.cpp"), os.path.join(src, "raytracer_host.cpp"),
                    os.path.join(src, "raytracer_math.cpp"), os.path.join(src, "webp_anim.cpp"), "-pthread",
                    "-L", os.path.join(REPO, "raytracer_amd", "csrc"), "-lrt_mi355", "-ldl", "-lm",
                    "-Wl,-rpath," + os.path.join(REPO, "raytracer_amd", "csrc")], check=True, timeout=600)
    r = subprocess.run([exe_t, host.ASSETS, str(scratch)], capture_output=True, text=True, timeout=900)
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("the thread sanitizer cannot map its shadow memory on this kernel")
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == '{"errors": 0}'


def test_no_gpu_means_an_error_not_a_cpu_render(capi, host):
    if gpu_present():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.RtError) as e:
        capi.Context(0)
    assert e.value.status == capi.RT_ERR_NO_DEVICE
    # null handles are rejected, nothing crashes
    L = capi.lib()
    assert L.rt_ctx_destroy(None) == capi.RT_ERR_INVALID
    assert L.rt_mesh_count(None) == capi.RT_ERR_INVALID
    assert L.rt_render_frame(None, None, None, None, 0, 8, 8, 0, None, None, None, None, None) == capi.RT_ERR_INVALID
    assert L.rt_render_frames(None, None, None, None, 1, 0, 8, 8, None, 192, None) == capi.RT_ERR_INVALID
    assert L.rt_render_frame_strips(None, None, None, None, 0, 8, 8, 2, 0, 0, None, None, None, None, None) == capi.RT_ERR_INVALID
    assert L.rt_mesh_upload_soup(None, None, 0, None, 1, 1, None, None) == capi.RT_ERR_INVALID
    assert L.rt_mesh_rebuild(None, 0, None, 0, None) == capi.RT_ERR_INVALID
    assert L.rt_bvh_build_device(None, None, 0, None, 0, None) == capi.RT_ERR_INVALID
    # page-locked memory needs the device runtime: the caller falls back to malloc (create_scene below does)
    p = C.c_void_p()
    assert L.rt_pinned_alloc(4096, C.byref(p)) == capi.RT_ERR_NO_DEVICE and not p.value
    assert L.rt_pinned_free(None) == 0
    # the reference API: render_scene reports the failure and leaves the frame untouched
    H = host.lib()
    scene = H.create_scene(64, 48, 1000, 4, 1.0)
    H.set_scene_camera(C.byref(scene), host.Vec3(-3, 3, -3), host.Vec3(0, 0, 0), host.Vec3(0, 1, 0), 60.0)
    H.set_scene_light(C.byref(scene), host.Vec3(1, 1, -1), host.Vec3(1.4, 1.4, 1.4))
    before = host.frame_array(scene, 0)
    H.render_scene(C.byref(scene))
    assert H.rt_host_last_status(C.byref(scene)) == capi.RT_ERR_NO_DEVICE
    assert np.array_equal(host.frame_array(scene, 0), before)
    H.destroy_scene(C.byref(scene))


# ------------------------------------------------------------------------------ host-only entry points

@pytest.mark.parametrize("name", NAMES)
def test_rt_bvh_build_equals_reference(capi, po, golden_bvh, name):
    soup = po.load_obj(os.path.join(po.ASSETS, name + ".obj"))
    post, nodes = capi.build_bvh(soup)
    assert np.array_equal(post.view(np.uint32), soup[golden_bvh[name + "_perm"]].view(np.uint32))
    want = golden_bvh[name + "_nodes"]
    assert nodes.shape == want.shape
    for f in ("left", "right", "start", "count"):
        assert np.array_equal(nodes[f], want[f])
    assert np.array_equal(bits(nodes["bmin"]), bits(want["bmin"])) and np.array_equal(bits(nodes["bmax"]), bits(want["bmax"]))


def signed_zero_soup(n, seed):
    """Triangles on a coarse grid with many coordinates that are exactly zero, of either sign."""
    rng = np.random.default_rng(seed)
    soup = np.zeros((n, 24), np.float32)
    v = rng.integers(-2, 3, size=(n, 9)).astype(np.float32) * np.float32(0.5)
    v = np.where((v == 0) & (rng.random((n, 9)) < 0.5), np.float32(-0.0), v)
    soup[:, :9] = v
    soup[:, 9:] = rng.random((n, 15), dtype=np.float32)
    return soup


def test_rt_bvh_build_keeps_the_reference_sign_of_zero_bounds(capi, po):
    """Box bounds are stored results of fminf/fmaxf (aabb.c:10-23): when +0 and -0 meet, the C library the
    reference is built with returns the later point's zero.  Flat, axis-aligned geometry at 0 hits this."""
    soup = signed_zero_soup(300, 5)
    post, nodes = capi.build_bvh(soup)
    want_t, want_n = po.build_bvh(soup)
    assert np.array_equal(bits(post), bits(want_t))
    assert nodes.tobytes() == want_n.tobytes()
    assert (bits(nodes["bmin"]) == 0x80000000).any() and (bits(nodes["bmin"]) == 0).any()
    if po.ref_available():
        R = po.RefScene([soup], [np.zeros((2, 2, 4), np.uint8)])
        assert R.nodes(0).tobytes() == nodes.tobytes() and np.array_equal(bits(R.triangles(0)), bits(post))
        R.close()


def test_rt_bvh_build_on_nan_infinite_and_overflowing_vertices(capi, po):
    """tests/adversarial.py::nonfinite_soups: rt_bvh_build against the oracle's build and, where it is present, the compiled
    reference's create_bvh -- triangle order and node array bit for bit."""
    import adversarial
    n = 0
    for name, soup in adversarial.nonfinite_soups(sizes=(7, 60, 500)):
        post, nodes = capi.build_bvh(soup)
        want_t, want_n = po.build_bvh(soup)
        assert np.array_equal(bits(post), bits(want_t)) and nodes.tobytes() == want_n.tobytes(), name
        if po.ref_available():
            R = po.RefScene([soup], [np.zeros((2, 2, 4), np.uint8)])
            assert R.nodes(0).tobytes() == nodes.tobytes() and np.array_equal(bits(R.triangles(0)), bits(post)), name
            R.close()
        n += 1
    assert n >= 30


def test_rt_bvh_build_argument_errors(capi):
    L = capi.lib()
    used = C.c_int32(0)
    tris = np.zeros((10, 24), np.float32)
    tris[:, 0] = np.arange(10)
    nodes = np.zeros(2, capi.NODE_DTYPE)
    assert L.rt_bvh_build(tris.ctypes.data, 10, nodes.ctypes.data, 2, C.byref(used)) == capi.RT_ERR_CAPACITY
    assert L.rt_bvh_build(None, 10, nodes.ctypes.data, 2, C.byref(used)) == capi.RT_ERR_INVALID
    assert L.rt_bvh_build(tris.ctypes.data, -1, nodes.ctypes.data, 2, C.byref(used)) == capi.RT_ERR_INVALID
    assert L.rt_bvh_build(tris.ctypes.data, 10, nodes.ctypes.data, 0, C.byref(used)) == capi.RT_ERR_INVALID
    # no triangles: one empty node, as create_bvh_node(tris, 0, 0) gives
    assert L.rt_bvh_build(None, 0, nodes.ctypes.data, 2, C.byref(used)) == 0 and used.value == 1
    assert nodes["count"][0] == 0 and nodes["left"][0] == -1


@pytest.mark.parametrize("name", NAMES)
def test_create_mesh_and_flatten_tree_equal_reference(capi, host, po, golden_bvh, golden_hashes, name):
    """create_mesh (mesh.c:6-119) through the host library: OBJ parse, BVH build in place, texture decode;
    then rt_bvh_flatten_tree on the pointer tree it leaves in Mesh.bvh.root."""
    import hashlib
    H = host.lib()
    m = H.create_mesh(os.path.join(host.ASSETS, name + ".obj").encode(), os.path.join(host.ASSETS, name + ".webp").encode())
    try:
        assert int(m.triangle_count) == golden_hashes["meshes"][name]["triangles"]
        tris = host.mesh_triangles(m)
        assert hashlib.sha256(tris.tobytes()).hexdigest() == golden_hashes["meshes"][name]["triangles_post_sha256"]
        tex = host.mesh_texture(m)
        assert hashlib.sha256(tex.tobytes()).hexdigest() == golden_hashes["textures"][name]["sha256"]
        want = golden_bvh[name + "_nodes"]
        flat = np.zeros(want.shape[0] + 4, capi.NODE_DTYPE)
        used = C.c_int32(0)
        st = capi.lib().rt_bvh_flatten_tree(C.cast(m.bvh.root, C.c_void_p), flat.ctypes.data, flat.shape[0], C.byref(used))
        assert st == 0 and used.value == want.shape[0]
        flat = flat[:used.value]
        for f in ("left", "right", "start", "count"):
            assert np.array_equal(flat[f], want[f])
        assert np.array_equal(bits(flat["bmin"]), bits(want["bmin"])) and np.array_equal(bits(flat["bmax"]), bits(want["bmax"]))
        # too small an output array is reported, not overrun
        small = np.zeros(3, capi.NODE_DTYPE)
        assert capi.lib().rt_bvh_flatten_tree(C.cast(m.bvh.root, C.c_void_p), small.ctypes.data, 3, C.byref(used)) == capi.RT_ERR_CAPACITY
        # sample_mesh_texture (mesh.c:136-150) against the oracle's restatement
        om = po.HostMesh(tris, want, tex).as_struct(po.Transform(po.vec3(0, 0, 0), po.vec3(0, 0, 0)))
        rng = np.random.default_rng(4)
        for u, v in np.concatenate([rng.uniform(-1.5, 2.5, size=(200, 2)), [[0, 0], [1, 1], [0.5, 1.0], [1.0, 0.0]]]):
            rgb = (C.c_float * 3)()
            po.lib().orc_sample_texture(C.byref(om), float(u), float(v), rgb)
            got = H.sample_mesh_texture(C.byref(m), float(u), float(v))
            assert bits(np.array([got.x, got.y, got.z])).tolist() == bits(np.array(list(rgb))).tolist()
    finally:
        H.destroy_mesh(C.byref(m))
    assert not m.triangles and not m.texture_data and not m.bvh.root      # pointers cleared: a second destroy is harmless
    H.destroy_mesh(C.byref(m))


def test_create_mesh_missing_files(host):
    H = host.lib()
    m = H.create_mesh(b"/nonexistent/a.obj", b"/nonexistent/a.webp")
    assert int(m.triangle_count) == 0 and not m.texture_data
    H.destroy_mesh(C.byref(m))


def test_scene_bookkeeping_matches_reference(host):
    """create_scene scales and allocates (scene.c:8-29); next_frame stops at the last frame (scene.c:48-50)."""
    H = host.lib()
    s = H.create_scene(800, 600, 4000, 24, 0.9)
    assert (s.width, s.height, s.frame_count, s.current_frame) == (720, 540, 96, 0)
    assert s.duration_ms == 4000 and s.fps == 24 and abs(s.scale_factor - 0.9) < 1e-7
    for _ in range(200):
        H.next_frame(C.byref(s))
    assert s.current_frame == 95
    cam = H.create_camera(host.Vec3(1, 2, 3), host.Vec3(0, 0, 0), host.Vec3(0, 1, 0), 45.0)
    assert (cam.position.x, cam.position.y, cam.position.z, cam.fov) == (1, 2, 3, 45.0)
    lit = H.create_directional_light(host.Vec3(1, 1, -1), host.Vec3(1.4, 1.4, 1.4))
    n = np.float32(1) / np.sqrt(np.float32(3))
    assert abs(lit.direction.x - n) < 1e-6 and lit.direction.z < 0
    H.destroy_scene(C.byref(s))


def test_demo_animation_equals_reference_driver(host, po):
    """Per-frame placements of raytracer.c:40-53."""
    H = host.lib()
    for f in (0, 1, 17, 45, 95):
        d, t = host.Transform(), host.Transform()
        H.rt_host_demo_animation(f, C.byref(d), C.byref(t))
        want = po.animation_transforms(f)
        for got, w in ((d, want[0]), (t, want[1])):
            g = np.array([got.position.x, got.position.y, got.position.z, got.rotation.x, got.rotation.y, got.rotation.z], np.float32)
            assert np.array_equal(bits(g), bits(np.array(w.position.tup() + w.rotation.tup(), np.float32)))


OBJ_MIXED = """# comment
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
f 1/1/1 2/2/1 3/3/1
f 1/1/1 2/2/1 3/3/1 4/4/1
f 1//1 2//1 5//1
f 1/1 2/2 5/3
f 1 4 5
f -5/-4/-1 -4/-3/-1 -1/-2/-1
f 1/1/1 2/2/1
f 1/1/1 2/2/1 9/3/1
""" + "f 1/1/1 2/2/1 3/3/1 " + " " * 5000 + "\n"


def test_obj_reader_modes(host, tmp_path):
    """Default mode reads what the reference reads (mesh.c:41-76: three v/vt/vn corners, the rest of the line ignored);
    the tolerant mode adds fans, optional vt/vn and negative indices.  Both accept long lines."""
    H = host.lib()
    obj = tmp_path / "mixed.obj"
    obj.write_text(OBJ_MIXED)
    tex = os.path.join(host.ASSETS, "ground.webp").encode()
    try:
        m = H.create_mesh(str(obj).encode(), tex)
        t = host.mesh_triangles(m)
        # full triangle, first three corners of the quad, the negative-index triangle (v1 v2 v5), the long line
        assert t.shape[0] == 4
        assert np.array_equal(t[0, 0:9], [0, 0, 0, 1, 0, 0, 1, 1, 0]) and np.array_equal(t[1], t[0])
        assert np.array_equal(t[2, 0:9], [0, 0, 0, 1, 0, 0, 0.5, 0.5, 1])
        assert np.array_equal(t[2, 9:15], [0, 0, 1, 0, 1, 1])            # vt -4 -3 -2 of four = 1 2 3
        H.destroy_mesh(C.byref(m))

        H.rt_host_set_obj_mode(1)
        m = H.create_mesh(str(obj).encode(), tex)
        t = host.mesh_triangles(m)
        assert t.shape[0] == 1 + 2 + 1 + 1 + 1 + 1 + 1                     # the two malformed faces are still skipped

        def rows(v9):                                                       # the BVH build reorders more than 4 triangles
            return [r for r in t if np.array_equal(r[0:9], np.array(v9, np.float32))]
        assert len(rows([0, 0, 0, 1, 0, 0, 1, 1, 0])) == 3                 # the triangle, the quad's first half, the long line
        assert len(rows([0, 0, 0, 1, 1, 0, 0, 1, 0])) == 1                 # second triangle of the fan: corners 1 3 4
        apex = rows([0, 0, 0, 1, 0, 0, 0.5, 0.5, 1])                       # v//vn, v/vt and the negative-index face
        assert len(apex) == 3
        n = np.cross(np.array([1, 0, 0.0]), np.array([0.5, 0.5, 1.0]))
        n /= np.linalg.norm(n)
        kinds = set()
        for r in apex:
            if np.array_equal(r[9:15], [0, 0, 0, 0, 0, 0]):               # v//vn: no texture coordinates, the file's normal
                assert np.array_equal(r[15:24], [0, 0, 1] * 3)
                kinds.add("v//vn")
            elif np.allclose(r[15:18], n, atol=1e-6):                     # v/vt: geometric normal
                assert np.array_equal(r[9:15], [0, 0, 1, 0, 1, 1])
                kinds.add("v/vt")
            else:                                                         # negative indices, everything given
                assert np.array_equal(r[9:15], [0, 0, 1, 0, 1, 1]) and np.array_equal(r[15:24], [0, 0, 1] * 3)
                kinds.add("negative")
        assert kinds == {"v//vn", "v/vt", "negative"}
        bare = rows([0, 0, 0, 0, 1, 0, 0.5, 0.5, 1])                       # "f 1 4 5"
        g = np.cross([0, 1, 0], [0.5, 0.5, 1.0])
        assert len(bare) == 1 and np.allclose(bare[0][15:24].reshape(3, 3), g / np.linalg.norm(g), atol=1e-6)
        assert m.bvh.root and int(m.bvh.triangle_count) == t.shape[0]
        H.destroy_mesh(C.byref(m))
    finally:
        H.rt_host_set_obj_mode(0)


def test_obj_reader_modes_agree_on_reference_assets(host, golden_hashes):
    import hashlib
    H = host.lib()
    try:
        for mode in (0, 1):
            H.rt_host_set_obj_mode(mode)
            m = H.create_mesh(os.path.join(host.ASSETS, "drone.obj").encode(), os.path.join(host.ASSETS, "drone.webp").encode())
            assert hashlib.sha256(host.mesh_triangles(m).tobytes()).hexdigest() == golden_hashes["meshes"]["drone"]["triangles_post_sha256"]
            H.destroy_mesh(C.byref(m))
    finally:
        H.rt_host_set_obj_mode(0)


def test_value_type_helpers_equal_reference_vectors(host, golden_kat, golden_frames):
    """The small functions of math/ray.h, geometry/aabb.h, render/camera.h and utils/image.h in the host library
    against the known-answer vectors taken from the compiled reference."""
    H = host.lib()
    g = golden_kat
    v3 = lambda a: host.Vec3(float(a[0]), float(a[1]), float(a[2]))          # noqa: E731
    n = 1024
    # ray_aabb_intersect (aabb.c:28-56)
    got = np.array([H.ray_aabb_intersect(host.Ray(v3(g["aabb_o"][k]), v3(g["aabb_d"][k])), host.AABB(v3(g["aabb_lo"][k]), v3(g["aabb_hi"][k])))
                    for k in range(g["aabb_hit"].shape[0])], np.int32)
    assert np.array_equal(got, g["aabb_hit"])
    # ray_triangle_intersect (ray.c:42-67)
    hit = np.zeros(g["tri_hit"].shape[0], np.int32)
    tuv = np.zeros((hit.size, 3), np.float32)
    for k in range(hit.size):
        t, u, v = C.c_float(0), C.c_float(0), C.c_float(0)
        hit[k] = H.ray_triangle_intersect(host.Ray(v3(g["tri_o"][k]), v3(g["tri_d"][k])), v3(g["tri_v"][k, 0]), v3(g["tri_v"][k, 1]),
                                          v3(g["tri_v"][k, 2]), C.byref(t), C.byref(u), C.byref(v))
        if hit[k]:
            tuv[k] = (t.value, u.value, v.value)
    assert np.array_equal(hit, g["tri_hit"]) and np.array_equal(bits(tuv), bits(g["tri_tuv"]))
    # transform_ray / transform_normal (ray.c:3-40)
    rays, nrm = np.zeros((n, 6), np.float32), np.zeros((n, 3), np.float32)
    for k in range(n):
        tf = host.Transform(v3(g["tf_pos"][k]), v3(g["tf_rot"][k]))
        r = H.transform_ray(host.Ray(v3(g["tf_o"][k]), v3(g["tf_d"][k])), tf)
        rays[k] = (r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z)
        m = H.transform_normal(v3(g["tf_d"][k]), tf)
        nrm[k] = (m.x, m.y, m.z)
    assert np.array_equal(bits(rays), bits(g["tf_ray"])) and np.array_equal(bits(nrm), bits(g["tf_normal"]))
    # get_camera_ray (camera.c:12-30)
    out = np.zeros((n, 6), np.float32)
    for k in range(n):
        p = g["cam_params"][k]
        cam = host.Camera(v3(p[0:3]), v3(p[3:6]), v3(p[6:9]), float(p[9]))
        r = H.get_camera_ray(C.byref(cam), float(g["cam_xy"][k, 0]), float(g["cam_xy"][k, 1]), float(g["cam_aspect"][k]))
        out[k] = (r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z)
    assert np.array_equal(bits(out), bits(g["cam_ray"]))
    # bicubic_interpolate (image.c:25-78)
    frame = np.ascontiguousarray(golden_frames["f0_rgb"])
    got = np.array([H.bicubic_interpolate(frame.ctypes.data, float(g["bic_x"][k]), float(g["bic_y"][k]), 240, 135)
                    for k in range(g["bic_x"].shape[0])], np.uint32)
    assert np.array_equal(got, g["bic_out"])
    assert H.get_pixel_rgb(frame.ctypes.data, -5, 999, 240, 135) == (int(frame[134, 0, 0]) << 16 | int(frame[134, 0, 1]) << 8 | int(frame[134, 0, 2]))


@pytest.mark.parametrize("name", NAMES)
def test_intersect_bvh_equals_reference_vectors(host, golden_kat, name):
    """intersect_bvh (accel/bvh.h:27-28, bvh.c:80-129) of the host library on the tree create_mesh builds: hit flag,
    t, u, v and triangle index of 2048 rays per mesh as the compiled reference returned them."""
    H = host.lib()
    g = golden_kat
    m = H.create_mesh(os.path.join(host.ASSETS, name + ".obj").encode(), os.path.join(host.ASSETS, name + ".webp").encode())
    try:
        n = g[f"bvh_{name}_hit"].shape[0]
        hit, tuv, idx = np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.full(n, -1, np.int32)
        v3 = lambda a: host.Vec3(float(a[0]), float(a[1]), float(a[2]))          # noqa: E731
        for k in range(n):
            t, u, v, i = C.c_float(1e30), C.c_float(-7), C.c_float(-7), C.c_int(-1)
            hit[k] = H.intersect_bvh(m.bvh.root, host.Ray(v3(g[f"bvh_{name}_o"][k]), v3(g[f"bvh_{name}_d"][k])), m.triangles,
                                     C.byref(t), C.byref(u), C.byref(v), C.byref(i))
            if hit[k]:
                tuv[k] = (t.value, u.value, v.value)
                idx[k] = i.value
            else:
                assert (t.value, u.value, v.value, i.value) == (np.float32(1e30), -7, -7, -1)      # outputs untouched on a miss
        assert np.array_equal(hit, g[f"bvh_{name}_hit"]) and np.array_equal(idx, g[f"bvh_{name}_idx"])
        assert np.array_equal(bits(tuv), bits(g[f"bvh_{name}_tuv"]))
        # a root that is a leaf honours the caller's bound (bvh.c:85); an inner root ignores it (bvh.c:106)
        k = int(np.flatnonzero(hit)[0])
        ray = host.Ray(v3(g[f"bvh_{name}_o"][k]), v3(g[f"bvh_{name}_d"][k]))
        t = C.c_float(tuv[k, 0] * 0.5)
        u, v, i = C.c_float(0), C.c_float(0), C.c_int(-1)
        assert H.intersect_bvh(m.bvh.root, ray, m.triangles, C.byref(t), C.byref(u), C.byref(v), C.byref(i)) and t.value == tuv[k, 0]
        tri = (host.Triangle * 1)(m.triangles[int(idx[k])])
        leaf = H.create_bvh(tri, 1)
        t = C.c_float(tuv[k, 0] * 0.5)
        assert not H.intersect_bvh(leaf.root, ray, tri, C.byref(t), C.byref(u), C.byref(v), C.byref(i))
        t = C.c_float(1e30)
        assert H.intersect_bvh(leaf.root, ray, tri, C.byref(t), C.byref(u), C.byref(v), C.byref(i)) and i.value == 0
        assert np.float32(t.value) == tuv[k, 0]
        H.destroy_bvh(C.byref(leaf))
    finally:
        H.destroy_mesh(C.byref(m))


def test_vec3_mat4_aabb_helpers(host):
    H = host.lib()
    a, b = host.Vec3(1, 2, 3), host.Vec3(-2, 0.5, 4)
    c = H.vec3_cross(a, b)
    assert (c.x, c.y, c.z) == (2 * 4 - 3 * 0.5, 3 * -2 - 1 * 4, 1 * 0.5 - 2 * -2)
    assert H.vec3_dot(a, b) == np.float32(1 * -2 + 2 * 0.5 + 3 * 4)
    nrm = H.vec3_normalize(a)
    inv_len = np.float32(1) / np.sqrt(np.float32(14))
    assert (nrm.x, nrm.y, nrm.z) == (np.float32(1) * inv_len, np.float32(2) * inv_len, np.float32(3) * inv_len)   # a * (1/len), vec3.c:15,35
    m = H.mat4_multiply(H.mat4_translation(host.Vec3(1, 2, 3)), H.mat4_rotation_y(0.7))
    back = H.mat4_multiply(m, H.mat4_inverse(m))
    assert np.allclose(np.array([[back.m[i][j] for j in range(4)] for i in range(4)]), np.eye(4), atol=1e-6)
    p = H.mat4_transform_point(m, host.Vec3(0, 0, 0))
    assert (p.x, p.y, p.z) == (1, 2, 3)
    t = H.mat4_transpose(m)
    assert t.m[3][0] == m.m[0][3] == 1
    e = H.create_empty_aabb()
    assert e.min.x == np.float32(1e30) and e.max.z == np.float32(-1e30)
    tri = host.Triangle(host.Vec3(0, 0, 0), host.Vec3(1, -1, 2), host.Vec3(-3, 4, 1))
    bb = H.get_triangle_bounds(tri)
    assert (bb.min.x, bb.min.y, bb.min.z, bb.max.x, bb.max.y, bb.max.z) == (-3, -1, 0, 1, 4, 2)
    assert H.cubic_hermite(1, 2, 3, 4, 0.0) == 2 and H.cubic_hermite(1, 2, 3, 4, 1.0) == 3
