"""ctypes front-end of the CPU oracle (oracle/liboracle.so) and of the compiled reference
(oracle/_ref/libref.so, present only when it was built where /root/reference exists).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
ASSETS = os.path.join(REPO, "raytracer_amd", "assets")

# ----------------------------------------------------------------------------- plain structs


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tup(self):
        return (self.x, self.y, self.z)


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class Transform(C.Structure):
    _fields_ = [("position", Vec3), ("rotation", Vec3)]


class Aabb(C.Structure):
    _fields_ = [("min", Vec3), ("max", Vec3)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("look_at", Vec3), ("up", Vec3), ("fov", C.c_float)]


class Light(C.Structure):
    _fields_ = [("direction", Vec3), ("color", Vec3)]


class Mesh(C.Structure):
    _fields_ = [
        ("triangles", C.c_void_p),
        ("triangle_count", C.c_int32),
        ("nodes", C.c_void_p),
        ("node_count", C.c_int32),
        ("texture_rgba", C.c_void_p),
        ("texture_width", C.c_int32),
        ("texture_height", C.c_int32),
        ("transform", Transform),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "shadow_rays", "slab_tests", "tri_tests", "hit_pixels", "shadowed_pixels")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# One triangle = 24 floats: v0 v1 v2 (9) | t0 t1 t2 (6) | n0 n1 n2 (9)   (96 bytes)
TRI_FLOATS = 24
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("left", "<i4"), ("right", "<i4"),
                       ("start", "<i4"), ("count", "<i4")])
assert NODE_DTYPE.itemsize == 40


def vec3(x, y, z) -> Vec3:
    return Vec3(float(x), float(y), float(z))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- library loading

_LIB = None
_REF = None


def lib():
    """The restatement.  Raises if oracle/liboracle.so has not been built."""
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_camera_ray.restype = Ray
        L.orc_camera_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float]
        L.orc_transform_ray.restype = Ray
        L.orc_transform_ray.argtypes = [Ray, Transform]
        L.orc_transform_normal.restype = Vec3
        L.orc_transform_normal.argtypes = [Vec3, Transform]
        L.orc_ray_aabb.restype = C.c_int
        L.orc_ray_aabb.argtypes = [Ray, Aabb]
        L.orc_ray_triangle.restype = C.c_int
        L.orc_ray_triangle.argtypes = [Ray, Vec3, Vec3, Vec3] + [C.POINTER(C.c_float)] * 3
        L.orc_intersect_bvh.restype = C.c_int
        L.orc_intersect_bvh.argtypes = [C.c_void_p, C.c_int32, Ray, C.c_void_p] + [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_int32)]
        L.orc_make_light.restype = Light
        L.orc_make_light.argtypes = [Vec3, Vec3]
        L.orc_sample_texture.restype = None
        L.orc_sample_texture.argtypes = [C.POINTER(Mesh), C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_bicubic.restype = C.c_uint32
        L.orc_bicubic.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int]
        L.orc_scaled_size.restype = None
        L.orc_scaled_size.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_upscale_frame.restype = None
        L.orc_upscale_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int32]
        L.orc_mesh_matrices.restype = None
        L.orc_mesh_matrices.argtypes = [Transform, C.c_void_p, C.c_void_p]
        L.orc_build_bvh.restype = C.c_int32
        L.orc_build_bvh.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.orc_load_obj.restype = C.c_int32
        L.orc_load_obj.argtypes = [C.c_char_p, C.c_void_p, C.c_int32] + [C.POINTER(C.c_int32)] * 3
        L.orc_render_frame_ids.restype = None
        L.orc_render_frame_ids.argtypes = [C.POINTER(Mesh), C.c_int32, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [
            C.POINTER(Counters), C.c_int32]
        L.orc_render_frame.restype = None
        L.orc_render_frame.argtypes = [C.POINTER(Mesh), C.c_int32, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int32]
        L.orc_sample_offset.restype = None
        L.orc_sample_offset.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_render_frame_sampled.restype = None
        L.orc_render_frame_sampled.argtypes = [C.POINTER(Mesh), C.c_int32, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.c_int32,
                                               C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int32]
        L.orc_render_window_sampled.restype = None
        L.orc_render_window_sampled.argtypes = [C.POINTER(Mesh), C.c_int32, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.c_int32,
                                                C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int32]
        L.orc_animation_transforms.restype = None
        L.orc_animation_transforms.argtypes = [C.c_int32, C.POINTER(Transform)]
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def ref_available() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))


def ref():
    """The compiled reference functions + composition loop (oracle/_ref/libref.so)."""
    global _REF
    if _REF is None:
        path = os.path.join(HERE, "_ref", "libref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libref.so missing (built only where /root/reference exists)")
        R = C.CDLL(path)
        R.refc_mesh_create.restype = C.c_void_p
        R.refc_mesh_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        R.refc_mesh_destroy.restype = None
        R.refc_mesh_destroy.argtypes = [C.c_void_p]
        R.refc_mesh_triangles.restype = None
        R.refc_mesh_triangles.argtypes = [C.c_void_p, C.c_void_p]
        R.refc_mesh_nodes.restype = C.c_int32
        R.refc_mesh_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        R.refc_camera_ray.restype = None
        R.refc_camera_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.POINTER(Ray)]
        R.refc_transform_ray.restype = None
        R.refc_transform_ray.argtypes = [C.POINTER(Ray), C.POINTER(Transform), C.POINTER(Ray)]
        R.refc_transform_normal.restype = None
        R.refc_transform_normal.argtypes = [C.POINTER(Vec3), C.POINTER(Transform), C.POINTER(Vec3)]
        R.refc_ray_aabb.restype = C.c_int
        R.refc_ray_aabb.argtypes = [C.POINTER(Ray), C.POINTER(Aabb)]
        R.refc_ray_triangle.restype = C.c_int
        R.refc_ray_triangle.argtypes = [C.POINTER(Ray)] + [C.POINTER(Vec3)] * 3 + [C.POINTER(C.c_float)] * 3
        R.refc_intersect.restype = C.c_int
        R.refc_intersect.argtypes = [C.c_void_p, C.POINTER(Ray)] + [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_int32)]
        R.refc_make_light.restype = None
        R.refc_make_light.argtypes = [C.POINTER(Vec3), C.POINTER(Vec3), C.POINTER(Light)]
        R.refc_bicubic.restype = C.c_uint32
        R.refc_bicubic.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int]
        R.refc_upscale_frame.restype = None
        R.refc_upscale_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        R.refc_render_frame.restype = None
        R.refc_render_frame.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Transform), C.c_int, C.POINTER(Camera), C.POINTER(Light),
                                        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        R.refc_max_threads.restype = C.c_int
        _REF = R
    return _REF


# ----------------------------------------------------------------------------- assets

def decode_webp(path: str) -> np.ndarray:
    """RGBA8 [h, w, 4] exactly as WebPDecodeRGBA returns it (mesh.c:105).  Uses the system
    libwebp when it can be loaded, Pillow otherwise (the two decode these files identically)."""
    data = open(path, "rb").read()
    try:
        W = C.CDLL("libwebp.so.7")
        W.WebPDecodeRGBA.restype = C.POINTER(C.c_uint8)
        W.WebPDecodeRGBA.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        W.WebPFree.argtypes = [C.c_void_p]
        w, h = C.c_int(0), C.c_int(0)
        p = W.WebPDecodeRGBA(data, len(data), C.byref(w), C.byref(h))
        if not p:
            raise OSError("WebPDecodeRGBA failed")
        out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
        W.WebPFree(p)
        return out
    except OSError:
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8))


def load_obj(path: str, cap: int = 1_000_000) -> np.ndarray:
    """Triangles [n, 24] float32 in file order (before any BVH build)."""
    buf = np.zeros((cap, TRI_FLOATS), dtype=np.float32)
    nv, nt, nn = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    n = lib().orc_load_obj(path.encode(), _ptr(buf), cap, C.byref(nv), C.byref(nt), C.byref(nn))
    if n < 0:
        raise FileNotFoundError(path)
    return buf[:n].copy()


def build_bvh(tris: np.ndarray):
    """Returns (triangles in post-build order, nodes) using the restated builder."""
    t = np.ascontiguousarray(tris, dtype=np.float32).copy()
    n = t.shape[0]
    nodes = np.zeros(max(2 * n, 1), dtype=NODE_DTYPE)
    used = lib().orc_build_bvh(_ptr(t), n, _ptr(nodes), nodes.shape[0])
    assert used > 0
    return t, nodes[:used].copy()


@dataclass
class HostMesh:
    tris: np.ndarray            # [n,24] f32, post-build order
    nodes: np.ndarray           # NODE_DTYPE
    rgba: np.ndarray            # [h,w,4] u8
    name: str = ""

    def as_struct(self, tf: Transform) -> Mesh:
        return Mesh(_ptr(self.tris), self.tris.shape[0], _ptr(self.nodes), self.nodes.shape[0],
                    _ptr(self.rgba), self.rgba.shape[1], self.rgba.shape[0], tf)


def load_mesh(name: str) -> HostMesh:
    tris, nodes = build_bvh(load_obj(os.path.join(ASSETS, name + ".obj")))
    return HostMesh(tris, nodes, decode_webp(os.path.join(ASSETS, name + ".webp")), name)


_SCENE_CACHE = None


def default_meshes() -> List[HostMesh]:
    """drone, treasure, ground -- the order of raytracer.c:27-34."""
    global _SCENE_CACHE
    if _SCENE_CACHE is None:
        _SCENE_CACHE = [load_mesh(n) for n in ("drone", "treasure", "ground")]
    return _SCENE_CACHE


def default_camera() -> Camera:                      # raytracer.c:13-18
    return Camera(vec3(-3, 3, -3), vec3(0, 0, 0), vec3(0, 1, 0), 60.0)


def default_light() -> Light:                        # raytracer.c:21-24 via light.c:3-5
    return lib().orc_make_light(vec3(1, 1, -1), vec3(1.4, 1.4, 1.4))


def animation_transforms(frame: int):
    arr = (Transform * 3)()
    lib().orc_animation_transforms(frame, arr)
    return [Transform(Vec3(*t.position.tup()), Vec3(*t.rotation.tup())) for t in arr]


def mesh_matrices(tf: Transform):
    inv = np.zeros(16, np.float32)
    nrm = np.zeros(16, np.float32)
    lib().orc_mesh_matrices(tf, _ptr(inv), _ptr(nrm))
    return inv.reshape(4, 4), nrm.reshape(4, 4)


# ----------------------------------------------------------------------------- frames

@dataclass
class Frame:
    rgb: np.ndarray            # [h,w,3] u8
    hit_id: np.ndarray         # [h,w] i32   (mesh<<24)|tri, -1 = miss
    hit_t: np.ndarray          # [h,w] f32
    shadow: np.ndarray         # [h,w] u8    1 blocked, 0 lit, 255 no hit
    counters: Optional[dict] = None


def render_frame(meshes: Sequence[HostMesh], transforms: Sequence[Transform], cam: Camera, light: Light,
                 width: int, height: int, threads: int = 0, debug: bool = True) -> Frame:
    arr = (Mesh * len(meshes))(*[m.as_struct(tf) for m, tf in zip(meshes, transforms)])
    rgb = np.zeros((height, width, 3), np.uint8)
    hid = np.zeros((height, width), np.int32)
    ht = np.zeros((height, width), np.float32)
    sh = np.zeros((height, width), np.uint8)
    cnt = Counters()
    lib().orc_render_frame(arr, len(meshes), C.byref(cam), C.byref(light), width, height, _ptr(rgb),
                           _ptr(hid) if debug else None, _ptr(ht) if debug else None, _ptr(sh) if debug else None,
                           C.byref(cnt), threads)
    return Frame(rgb, hid, ht, sh, cnt.as_dict())


def render_frame_ids(meshes: Sequence[HostMesh], transforms: Sequence[Transform], cam: Camera, light: Light,
                     width: int, height: int, threads: int = 0):
    """render_frame for mesh lists of any length: returns (Frame with hit_id = None, hit_mesh [h,w] i32, hit_tri [h,w] i32)."""
    arr = (Mesh * len(meshes))(*[m.as_struct(tf) for m, tf in zip(meshes, transforms)])
    rgb = np.zeros((height, width, 3), np.uint8)
    hm = np.zeros((height, width), np.int32)
    htri = np.zeros((height, width), np.int32)
    ht = np.zeros((height, width), np.float32)
    sh = np.zeros((height, width), np.uint8)
    cnt = Counters()
    lib().orc_render_frame_ids(arr, len(meshes), C.byref(cam), C.byref(light), width, height, _ptr(rgb), _ptr(hm), _ptr(htri), _ptr(ht),
                               _ptr(sh), C.byref(cnt), threads)
    return Frame(rgb, None, ht, sh, cnt.as_dict()), hm, htri


def render_frame_sampled(meshes: Sequence[HostMesh], transforms: Sequence[Transform], cam: Camera, lights: Sequence[Light],
                         spp: int, seed: int, width: int, height: int, threads: int = 0) -> Frame:
    """Several camera rays per pixel, several lights (rt_render_frame_sampled's definition).  hit_id is sample 0's."""
    arr = (Mesh * len(meshes))(*[m.as_struct(tf) for m, tf in zip(meshes, transforms)])
    larr = (Light * len(lights))(*lights)
    rgb = np.zeros((height, width, 3), np.uint8)
    hid = np.zeros((height, width), np.int32)
    cnt = Counters()
    lib().orc_render_frame_sampled(arr, len(meshes), C.byref(cam), larr, len(lights), spp, seed, width, height,
                                   _ptr(rgb), _ptr(hid), C.byref(cnt), threads)
    return Frame(rgb, hid, None, None, cnt.as_dict())


def render_window_sampled(meshes: Sequence[HostMesh], transforms: Sequence[Transform], cam: Camera, lights: Sequence[Light],
                          spp: int, seed: int, width: int, height: int, x0: int, y0: int, win_w: int, win_h: int, threads: int = 0) -> Frame:
    """The pixels [x0, x0 + win_w) x [y0, y0 + win_h) of render_frame_sampled's width x height frame, as planes of the window's size."""
    arr = (Mesh * len(meshes))(*[m.as_struct(tf) for m, tf in zip(meshes, transforms)])
    larr = (Light * len(lights))(*lights)
    rgb = np.zeros((win_h, win_w, 3), np.uint8)
    hid = np.zeros((win_h, win_w), np.int32)
    cnt = Counters()
    lib().orc_render_window_sampled(arr, len(meshes), C.byref(cam), larr, len(lights), spp, seed, width, height, x0, y0, win_w, win_h,
                                    _ptr(rgb), _ptr(hid), C.byref(cnt), threads)
    return Frame(rgb, hid, None, None, cnt.as_dict())


def scaled_size(width: int, height: int, scale_factor: float):
    """Output size of the encoder's upscale (scene.c:164-165)."""
    sw, sh = C.c_int(0), C.c_int(0)
    lib().orc_scaled_size(width, height, scale_factor, C.byref(sw), C.byref(sh))
    return sw.value, sh.value


def upscale_frame(rgb: np.ndarray, scaled_width: int, scaled_height: int, threads: int = 0, use_ref: bool = False) -> np.ndarray:
    """The bicubic upscale of one RGB8 frame to ARGB uint32 [scaled_height, scaled_width] (scene.c:188-202)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    out = np.zeros((scaled_height, scaled_width), np.uint32)
    fn = ref().refc_upscale_frame if use_ref else lib().orc_upscale_frame
    fn(_ptr(rgb), w, h, scaled_width, scaled_height, _ptr(out), threads)
    return out


class RefScene:
    """The compiled reference's BVHs over the same triangle soup (file order in, reference build)."""

    def __init__(self, soups: Sequence[np.ndarray], textures: Sequence[np.ndarray]):
        R = ref()
        self._keep = [(np.ascontiguousarray(s, np.float32), np.ascontiguousarray(t)) for s, t in zip(soups, textures)]
        self.handles = [R.refc_mesh_create(_ptr(s), s.shape[0], _ptr(t), t.shape[1], t.shape[0]) for s, t in self._keep]

    def triangles(self, i: int) -> np.ndarray:
        out = np.zeros_like(self._keep[i][0])
        ref().refc_mesh_triangles(self.handles[i], _ptr(out))
        return out

    def nodes(self, i: int) -> np.ndarray:
        n = self._keep[i][0].shape[0]
        out = np.zeros(max(2 * n, 1), NODE_DTYPE)
        used = ref().refc_mesh_nodes(self.handles[i], _ptr(out), out.shape[0])
        return out[:used].copy()

    def render(self, transforms, cam, light, width, height, threads=0) -> Frame:
        hs = (C.c_void_p * len(self.handles))(*self.handles)
        tfs = (Transform * len(transforms))(*transforms)
        rgb = np.zeros((height, width, 3), np.uint8)
        hid = np.zeros((height, width), np.int32)
        ht = np.zeros((height, width), np.float32)
        sh = np.zeros((height, width), np.uint8)
        ref().refc_render_frame(hs, tfs, len(self.handles), C.byref(cam), C.byref(light), width, height,
                                _ptr(rgb), _ptr(hid), _ptr(ht), _ptr(sh), threads)
        return Frame(rgb, hid, ht, sh)

    def close(self):
        for h in self.handles:
            ref().refc_mesh_destroy(h)
        self.handles = []
