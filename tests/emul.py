"""ctypes wrapper of tests/host_emul/libemul.so: the product's packer (rt_pack.h) and pixel
pipeline (rt_pixel.h) compiled for the host, so that they can be checked against the oracle on a
machine without a GPU.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "host_emul", "libemul.so")

_lib = None


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Transform(C.Structure):
    _fields_ = [("position", Vec3), ("rotation", Vec3)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("look_at", Vec3), ("up", Vec3), ("fov", C.c_float)]


class Light(C.Structure):
    _fields_ = [("direction", Vec3), ("color", Vec3)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "shadow_rays", "slab_tests", "tri_tests", "hit_pixels", "shadowed_pixels", "skip_checks",
                 "vis_disagreements", "shadow_vis_disagreements", "entry_disagreements", "cull_disagreements",
                 "instance_disagreements", "idle_shadow_rays")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            raise RuntimeError(f"{PATH} missing: run `make` at the repo root")
        L = C.CDLL(PATH)
        vp, i32 = C.c_void_p, C.c_int32
        L.emul_scene_create.restype = vp
        L.emul_scene_destroy.argtypes = [vp]
        L.emul_last_error.argtypes = [vp]
        L.emul_last_error.restype = C.c_char_p
        L.emul_stack_high.argtypes = [vp]
        L.emul_mesh_add.argtypes = [vp, vp, i32, vp, i32, vp, i32, i32]
        L.emul_mesh_info.argtypes = [vp, C.c_int, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_uint32)]
        L.emul_mesh_prios.argtypes = [vp, C.c_int, vp]
        L.emul_placement.argtypes = [C.POINTER(Transform), C.POINTER(Camera), C.POINTER(Light), vp, vp, vp, vp,
                                     C.POINTER(C.c_uint32)]
        L.emul_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Transform), i32, i32, i32,
                                  vp, vp, vp, vp, C.POINTER(Counters), i32, vp, vp]
        L.emul_screen_rect.argtypes = [C.POINTER(Transform), vp, vp, C.POINTER(Camera), C.POINTER(Light), i32, i32, vp]
        L.emul_screen_rect.restype = None
        L.emul_shadow_rect.argtypes = [C.POINTER(Transform), vp, vp, vp, vp, C.POINTER(Camera), C.POINTER(Light), i32, i32, vp]
        L.emul_shadow_rect.restype = None
        L.emul_sample_offset.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_float)] * 2
        L.emul_sample_offset.restype = None
        L.emul_render_sampled.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), i32, C.POINTER(Transform), i32, i32, C.c_uint32,
                                          i32, i32, vp, vp, C.POINTER(Counters), i32]
        L.emul_render_instanced.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), vp, vp, i32, i32, i32, vp, vp, vp, vp, vp,
                                            C.POINTER(Counters), C.POINTER(i32), C.POINTER(i32), i32]
        L.emul_upscale.argtypes = [vp, i32, i32, i32, i32, i32, vp]
        L.emul_upscale.restype = None
        L.emul_set_regroup.argtypes = [vp, i32]
        L.emul_set_regroup.restype = None
        L.emul_keep_hierarchy.argtypes = [vp, i32]
        L.emul_keep_hierarchy.restype = None
        L.emul_hierarchy_reused.argtypes = [vp]
        L.emul_regroup_routes_agree.argtypes = [vp, i32, vp, i32]
        L.emul_work_plane.argtypes = [vp, vp]
        L.emul_work_plane.restype = None
        L.emul_mesh_wide.argtypes = [vp, i32]
        L.emul_mesh_grid_axis.argtypes = [vp, i32]
        L.emul_damage_shear.argtypes = [vp, i32]
        L.emul_set_loose.argtypes = [vp, i32]
        L.emul_set_loose.restype = None
        L.emul_set_vis_shrink.argtypes = [vp, i32]
        L.emul_set_vis_shrink.restype = None
        L.emul_damage_grid.argtypes = [vp, i32]
        L.emul_damage_grid_masks.argtypes = [vp, i32]
        L.emul_damage_second_tree.argtypes = [vp, i32]
        _lib = L
    return _lib


def regroup_routes_agree(tris, nodes):
    """0: a mesh packed with its second tree equals the same mesh packed without and regrouped afterwards (the route of meshes
    packed on the device); 2: no second tree either way."""
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 24)
    nodes = np.ascontiguousarray(nodes)
    return lib().emul_regroup_routes_agree(tris.ctypes.data, tris.shape[0], nodes.ctypes.data, nodes.shape[0])


def upscale(rgb, scaled_width, scaled_height, rows=8):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    out = np.zeros((scaled_height, scaled_width), np.uint32)
    lib().emul_upscale(rgb.ctypes.data, rgb.shape[1], rgb.shape[0], scaled_width, scaled_height, rows, out.ctypes.data)
    return out


def screen_rect(tf, bmin, bmax, cam, light, width, height):
    """mesh_screen_rect: tiles {x0, y0, x1, y1} outside of which camera rays cannot pass the placed box."""
    lo = np.ascontiguousarray(bmin, np.float32)
    hi = np.ascontiguousarray(bmax, np.float32)
    out = np.zeros(4, np.int32)
    t, c, l = conv(tf, Transform), conv(cam, Camera), conv(light, Light)
    lib().emul_screen_rect(C.byref(t), lo.ctypes.data, hi.ctypes.data, C.byref(c), C.byref(l), width, height, out.ctypes.data)
    return [int(v) for v in out]


def shadow_rect(tf, bmin, bmax, scene_lo, scene_hi, cam, light, width, height):
    """mesh_shadow_rect: tiles outside of which no point of the scene's bounds sends a shadow ray through the placed box."""
    lo = np.ascontiguousarray(bmin, np.float32)
    hi = np.ascontiguousarray(bmax, np.float32)
    slo = np.ascontiguousarray(scene_lo, np.float64)
    shi = np.ascontiguousarray(scene_hi, np.float64)
    out = np.zeros(4, np.int32)
    t, c, l = conv(tf, Transform), conv(cam, Camera), conv(light, Light)
    lib().emul_shadow_rect(C.byref(t), lo.ctypes.data, hi.ctypes.data, slo.ctypes.data, shi.ctypes.data, C.byref(c), C.byref(l),
                           width, height, out.ctypes.data)
    return [int(v) for v in out]


def conv(obj, cls):
    """Copies an oracle ctypes struct (same layout) into this module's struct type."""
    out = cls()
    C.memmove(C.byref(out), C.byref(obj), C.sizeof(cls))
    return out


class Scene:
    def __init__(self, regroup=False):
        self.h = lib().emul_scene_create()
        lib().emul_set_regroup(self.h, 1 if regroup else 0)
        self.n = 0
        self._keep = []

    def add(self, tris, nodes, rgba) -> int:
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 24)
        nodes = np.ascontiguousarray(nodes)
        rgba = np.ascontiguousarray(rgba, np.uint8)
        self._keep.append((tris, nodes, rgba))
        st = lib().emul_mesh_add(self.h, tris.ctypes.data, tris.shape[0], nodes.ctypes.data, nodes.shape[0],
                                 rgba.ctypes.data, rgba.shape[1], rgba.shape[0])
        if st == 0:
            self.n += 1
        return st

    def error(self) -> str:
        return lib().emul_last_error(self.h).decode()

    def info(self, mesh):
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_uint32(0)
        assert lib().emul_mesh_info(self.h, mesh, C.byref(a), C.byref(b), C.byref(c)) == 0
        return {"pairs": a.value, "depth": b.value, "root_ref": c.value}

    def prios(self, mesh, n):
        out = np.zeros(n, np.uint32)
        assert lib().emul_mesh_prios(self.h, mesh, out.ctypes.data) == 0
        return out

    def render(self, cam, light, tfs, w, h, full=False, per_pixel=False, screen_cull=False, verify=False, colour_only=False):
        arr = (Transform * max(len(tfs), 1))(*[conv(t, Transform) for t in tfs])
        rgb = np.zeros((h, w, 3), np.uint8)
        hid = np.zeros((h, w), np.int32)
        ht = np.zeros((h, w), np.float32)
        sh = np.zeros((h, w), np.uint8)
        cnt = Counters()
        ps = np.zeros((h, w), np.uint32) if per_pixel else None
        pt = np.zeros((h, w), np.uint32) if per_pixel else None
        cam2, light2 = conv(cam, Camera), conv(light, Light)
        st = lib().emul_render(self.h, C.byref(cam2), C.byref(light2), arr, len(tfs), w, h, rgb.ctypes.data,
                               hid.ctypes.data, ht.ctypes.data, sh.ctypes.data, C.byref(cnt), 4 if colour_only else (3 if verify else (2 if screen_cull else (1 if full else 0))),
                               ps.ctypes.data if per_pixel else None, pt.ctypes.data if per_pixel else None)
        assert st == 0, self.error()
        return {"rgb": rgb, "hit_id": hid, "hit_t": ht, "shadow": sh, "counters": cnt.as_dict(), "px_slabs": ps,
                "px_tris": pt, "stack_high": lib().emul_stack_high(self.h)}

    def render_sampled(self, cam, lights, tfs, spp, seed, w, h, verify=False):
        arr = (Transform * max(len(tfs), 1))(*[conv(t, Transform) for t in tfs])
        larr = (Light * len(lights))(*[conv(l, Light) for l in lights])
        rgb = np.zeros((h, w, 3), np.uint8)
        hid = np.zeros((h, w), np.int32)
        cnt = Counters()
        cam2 = conv(cam, Camera)
        st = lib().emul_render_sampled(self.h, C.byref(cam2), larr, len(lights), arr, len(tfs), spp, seed, w, h,
                                       rgb.ctypes.data, hid.ctypes.data, C.byref(cnt), 1 if verify else 0)
        assert st == 0, self.error()
        return {"rgb": rgb, "hit_id": hid, "counters": cnt.as_dict()}

    def render_instanced(self, cam, light, mesh_ids, tfs, w, h, verify=False):
        ids = np.ascontiguousarray(mesh_ids, np.int32)
        arr = (Transform * max(len(tfs), 1))(*[conv(t, Transform) for t in tfs])
        rgb = np.zeros((h, w, 3), np.uint8)
        inst = np.zeros((h, w), np.int32)
        tri = np.zeros((h, w), np.int32)
        ht = np.zeros((h, w), np.float32)
        sh = np.zeros((h, w), np.uint8)
        cnt = Counters()
        pairs, depth = C.c_int32(0), C.c_int32(0)
        cam2, light2 = conv(cam, Camera), conv(light, Light)
        st = lib().emul_render_instanced(self.h, C.byref(cam2), C.byref(light2), ids.ctypes.data, arr, len(tfs), w, h,
                                         rgb.ctypes.data, inst.ctypes.data, tri.ctypes.data, ht.ctypes.data, sh.ctypes.data,
                                         C.byref(cnt), C.byref(pairs), C.byref(depth), 1 if verify else 0)
        assert st == 0, self.error()
        return {"rgb": rgb, "hit_inst": inst, "hit_tri": tri, "hit_t": ht, "shadow": sh, "counters": cnt.as_dict(),
                "tlas_pairs": pairs.value, "tlas_depth": depth.value}

    def wide(self, mesh: int) -> bool:
        """Whether the packer marked the mesh as holding a wide triangle (walked without the distance cull)."""
        r = lib().emul_mesh_wide(self.h, int(mesh))
        assert r >= 0
        return bool(r)

    def grid_axis(self, mesh: int) -> int:
        """0..2: flat mesh with a grid; 3: neither; 4 + k: a tree in the sheared frame that keeps axis k."""
        r = lib().emul_mesh_grid_axis(self.h, int(mesh))
        assert r >= 0
        return r

    def shear_masks_hold(self, mesh, cam, light, tf, w, h):
        """(rays for which the host's per-frame masks miss a family the gate itself finds close, rays compared, families spared)."""
        looked, spared = C.c_longlong(0), C.c_longlong(0)
        cam2, light2, tf2 = conv(cam, Camera), conv(light, Light), conv(tf, Transform)
        L = lib()
        L.emul_shear_masks_hold.restype = C.c_longlong
        L.emul_shear_masks_hold.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Transform), C.c_int32, C.c_int32,
                                            C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        bad = L.emul_shear_masks_hold(self.h, int(mesh), C.byref(cam2), C.byref(light2), C.byref(tf2), int(w), int(h), C.byref(looked), C.byref(spared))
        return int(bad), looked.value, spared.value

    def damage_shear(self, mesh: int) -> int:
        return lib().emul_damage_shear(self.h, int(mesh))

    def set_loose(self, on: bool):
        """Rays enter every mesh through its tree of single triangles (what the kernels' fast build walks); not bit-exact."""
        lib().emul_set_loose(self.h, 1 if on else 0)

    # deliberate damage, for the tests that the self-checking party (verify=True) notices wrong shortcuts
    def set_vis_shrink(self, tiles: int):
        lib().emul_set_vis_shrink(self.h, int(tiles))

    def damage_grid(self, mesh: int) -> int:
        return lib().emul_damage_grid(self.h, int(mesh))

    def damage_grid_masks(self, mesh: int) -> int:
        return lib().emul_damage_grid_masks(self.h, int(mesh))

    def damage_second_tree(self, mesh: int) -> int:
        return lib().emul_damage_second_tree(self.h, int(mesh))

    def record_work(self, plane):
        """plane: uint32 array [h, w, 2] that instanced frames fill with each pixel's {box tests, triangle tests}; None stops it.
        The caller keeps the array alive while frames are rendered."""
        lib().emul_work_plane(self.h, plane.ctypes.data if plane is not None else None)

    def keep_hierarchy(self, on: bool):
        """Instanced frames from now on keep the hierarchy over the instances between calls, as the product does."""
        lib().emul_keep_hierarchy(self.h, 1 if on else 0)

    def hierarchy_reused(self) -> int:
        return lib().emul_hierarchy_reused(self.h)

    def close(self):
        if self.h:
            lib().emul_scene_destroy(self.h)
            self.h = None
