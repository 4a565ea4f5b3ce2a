"""ctypes binding of the C ABI in include/rt_mi355.h (raytracer_amd/csrc/librt_mi355.so).

The library is the product: if it has not been built this module raises at import of the
symbols, and every rendering call fails loudly when no GPU is present.  Nothing here falls
back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_mi355.so")

RT_OK = 0
RT_ERR_INVALID, RT_ERR_DEVICE, RT_ERR_NO_DEVICE, RT_ERR_UNSUPPORTED, RT_ERR_CAPACITY = -1, -2, -3, -4, -5
RT_FLAG_COUNT_WORK = 0x1
RT_FLAG_REFERENCE_ORDER = 0x2
RT_FLAG_VERIFY_SKIPS = 0x4
RT_MAX_MESHES = 8
RT_MAX_BATCH = 16
RT_MAX_LIGHTS = 4
RT_MAX_SAMPLES = 64


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Transform(C.Structure):
    _fields_ = [("position", Vec3), ("rotation", Vec3)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("look_at", Vec3), ("up", Vec3), ("fov", C.c_float)]


class Light(C.Structure):
    _fields_ = [("direction", Vec3), ("color", Vec3)]


class Instance(C.Structure):
    _fields_ = [("mesh", C.c_int32), ("transform", Transform)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "shadow_rays", "slab_tests", "tri_tests", "hit_pixels", "shadowed_pixels", "skip_checks",
                 "vis_disagreements", "shadow_vis_disagreements", "entry_disagreements", "cull_disagreements",
                 "instance_disagreements", "idle_shadow_rays")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


TRI_FLOATS = 24          # rt_triangle: 96 bytes
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("left", "<i4"), ("right", "<i4"),
                       ("start", "<i4"), ("count", "<i4")])     # rt_flat_node: 40 bytes

EXPORTS = [
    "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_status_string", "rt_pinned_alloc", "rt_pinned_free", "rt_strip_rows", "rt_render_frame_strips", "rt_render_frame_sampled_strips", "rt_mesh_replace", "rt_mesh_upload_soup", "rt_mesh_rebuild", "rt_bvh_build", "rt_bvh_build_device", "rt_bvh_build_stats", "rt_bvh_flatten_tree",
    "rt_mesh_upload", "rt_mesh_upload_tree", "rt_scene_clear", "rt_mesh_count", "rt_render_frame",
    "rt_render_frame_host", "rt_sync", "rt_last_kernel_ms", "rt_last_counters", "rt_set_option", "rt_get_option",
    "rt_set_trace_buffer", "rt_render_frames", "rt_scaled_size", "rt_upscale_frames", "rt_upscale_frame_host",
    "rt_render_frame_sampled", "rt_render_frame_instanced", "rt_render_frame_instanced_host", "rt_render_frames_noalloc", "rt_render_frames_sharded", "rt_device_count",
]

_lib = None


class RtError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        super().__init__(f"rt_mi355 status {status} ({status_string(status)}){': ' + detail if detail else ''}")


def _load_torch_hip_first():
    """PyTorch wheels bundle their own HIP runtime (same soname as the system one).  A process that
    uses both torch and this library must let torch load its copy first: with the system runtime
    already mapped, torch's device initialisation fails ("No HIP GPUs are available")."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make product` (hipcc, gfx950); "
                               "there is no CPU fallback")
        _load_torch_hip_first()
        L = C.CDLL(LIB_PATH)
        vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
        L.rt_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rt_ctx_destroy.argtypes = [vp]
        L.rt_last_error.argtypes = [vp]
        L.rt_last_error.restype = C.c_char_p
        L.rt_status_string.argtypes = [C.c_int]
        L.rt_status_string.restype = C.c_char_p
        L.rt_bvh_build.argtypes = [vp, i32, vp, i32, C.POINTER(i32)]
        L.rt_bvh_build_device.argtypes = [vp, vp, i32, vp, i32, C.POINTER(i32)]
        L.rt_bvh_build_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32), C.POINTER(i32)]
        L.rt_bvh_flatten_tree.argtypes = [vp, vp, i32, C.POINTER(i32)]
        L.rt_mesh_upload.argtypes = [vp, vp, i32, vp, i32, vp, i32, i32, C.POINTER(i32)]
        L.rt_mesh_upload_tree.argtypes = [vp, vp, i32, vp, vp, i32, i32, C.POINTER(i32)]
        L.rt_mesh_replace.argtypes = [vp, i32, vp, i32, vp, i32]
        L.rt_mesh_upload_soup.argtypes = [vp, vp, i32, vp, i32, i32, C.POINTER(i32), vp]
        L.rt_mesh_rebuild.argtypes = [vp, i32, vp, i32, vp]
        L.rt_scene_clear.argtypes = [vp]
        L.rt_mesh_count.argtypes = [vp]
        L.rt_render_frame.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Transform), i32, i32, i32, u32,
                                      vp, vp, vp, vp, vp]
        L.rt_render_frame_host.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Transform), i32, i32, i32, u32,
                                           vp, vp, vp, vp]
        L.rt_render_frames.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Transform), i32, i32, i32, i32,
                                       vp, C.c_size_t, vp]
        L.rt_render_frames_noalloc.argtypes = L.rt_render_frames.argtypes
        L.rt_render_frames_sharded.argtypes = [C.POINTER(vp), i32, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Transform), i32, i32, i32, i32,
                                               vp, C.c_size_t]
        L.rt_device_count.argtypes = []
        L.rt_render_frame_sampled.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), i32, C.POINTER(Transform), i32, i32, i32, i32,
                                              u32, u32, vp, vp, vp]
        L.rt_strip_rows.argtypes = [i32, i32, i32]
        L.rt_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
        L.rt_pinned_free.argtypes = [vp]
        L.rt_render_frame_strips.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Transform), i32, i32, i32, i32, i32, u32,
                                             vp, vp, vp, vp, vp]
        L.rt_render_frame_sampled_strips.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), i32, C.POINTER(Transform), i32, i32, i32, i32,
                                                     u32, i32, i32, u32, vp, vp, vp]
        L.rt_render_frame_instanced.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Instance), i32, i32, i32, u32,
                                                vp, vp, vp, vp, vp, vp]
        L.rt_render_frame_instanced_host.argtypes = [vp, C.POINTER(Camera), C.POINTER(Light), C.POINTER(Instance), i32, i32, i32, u32, vp]
        L.rt_scaled_size.argtypes = [i32, i32, C.c_float, C.POINTER(i32), C.POINTER(i32)]
        L.rt_upscale_frames.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]
        L.rt_upscale_frame_host.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.rt_sync.argtypes = [vp, vp]
        L.rt_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.rt_last_counters.argtypes = [vp, C.POINTER(Counters)]
        L.rt_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        L.rt_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.rt_set_trace_buffer.argtypes = [vp, vp]
        for name in EXPORTS:
            getattr(L, name)          # AttributeError here = the library does not export the ABI
        _lib = L
    return _lib


def status_string(status: int) -> str:
    return lib().rt_status_string(status).decode()


def scaled_size(width: int, height: int, scale_factor: float):
    """rt_scaled_size: the upscaled size save_scene derives (scene.c:164-165)."""
    sw, sh = C.c_int32(0), C.c_int32(0)
    st = lib().rt_scaled_size(int(width), int(height), float(scale_factor), C.byref(sw), C.byref(sh))
    if st != RT_OK:
        raise RtError(st)
    return sw.value, sh.value


def strip_rows(height: int, strip_stride: int, strip_offset: int) -> int:
    """rt_strip_rows: rows of a frame that belong to strips strip_offset, strip_offset + strip_stride, ..."""
    n = lib().rt_strip_rows(int(height), int(strip_stride), int(strip_offset))
    if n < 0:
        raise RtError(n)
    return n


def build_bvh(tris: np.ndarray):
    """rt_bvh_build on a copy: returns (triangles in post-build order [n,24] f32, nodes NODE_DTYPE)."""
    t = np.ascontiguousarray(tris, dtype=np.float32).copy()
    n = int(t.shape[0])
    nodes = np.zeros(max(2 * n, 1), dtype=NODE_DTYPE)
    used = C.c_int32(0)
    st = lib().rt_bvh_build(t.ctypes.data, n, nodes.ctypes.data, int(nodes.shape[0]), C.byref(used))
    if st != RT_OK:
        raise RtError(st)
    return t, nodes[:used.value].copy()


def device_count() -> int:
    return int(lib().rt_device_count())


def render_frames_sharded(contexts, camera: Camera, light: Light, transforms, width: int, height: int, rgb,
                          frame_stride: Optional[int] = None):
    """rt_render_frames_sharded: frame f by contexts[f mod N], all frames gathered in `rgb` ([n_frames, h, w, 3] uint8 on
    contexts[0]'s GPU).  transforms[f] are the placements of frame f.  Returns when the frames are there."""
    n = len(transforms)
    m = len(transforms[0]) if n else 0
    flat = [t for frame in transforms for t in frame]
    arr = (Transform * max(len(flat), 1))(*flat)
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    stride = int(frame_stride) if frame_stride is not None else int(width) * int(height) * 3
    st = lib().rt_render_frames_sharded(handles, len(contexts), C.byref(camera), C.byref(light), arr, n, m, int(width), int(height),
                                        _dev_ptr(rgb), stride)
    if st != RT_OK:
        raise RtError(st, lib().rt_last_error(contexts[0]._h).decode() if contexts else "")


def _dev_ptr(x) -> Optional[int]:
    """Device pointer of a torch tensor (or an int / None passed through)."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return int(x.data_ptr())


class Context:
    """One GPU rendering context (rt_ctx)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        st = lib().rt_ctx_create(int(device), C.byref(self._h))
        if st != RT_OK:
            self._h = C.c_void_p()
            raise RtError(st, "rt_ctx_create")
        self.device = int(device)

    def _check(self, st: int):
        if st != RT_OK:
            raise RtError(st, lib().rt_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().rt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_mesh(self, tris: np.ndarray, nodes: np.ndarray, rgba: np.ndarray) -> int:
        tris = np.ascontiguousarray(tris, dtype=np.float32)
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert tris.ndim == 2 and tris.shape[1] == TRI_FLOATS and rgba.ndim == 3 and rgba.shape[2] == 4
        mid = C.c_int32(-1)
        self._check(lib().rt_mesh_upload(self._h, tris.ctypes.data, int(tris.shape[0]), nodes.ctypes.data,
                                         int(nodes.shape[0]), rgba.ctypes.data, int(rgba.shape[1]), int(rgba.shape[0]),
                                         C.byref(mid)))
        return mid.value

    def replace_mesh(self, mesh_id: int, tris: np.ndarray, nodes: np.ndarray):
        """rt_mesh_replace: new triangles (post-build order) and BVH for an uploaded mesh; the texture stays."""
        tris = np.ascontiguousarray(tris, dtype=np.float32)
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        assert tris.ndim == 2 and tris.shape[1] == TRI_FLOATS
        self._check(lib().rt_mesh_replace(self._h, int(mesh_id), tris.ctypes.data, int(tris.shape[0]), nodes.ctypes.data, int(nodes.shape[0])))

    def upload_soup(self, tris: np.ndarray, rgba: np.ndarray, want_triangles: bool = False):
        """rt_mesh_upload_soup: triangles in file order -> mesh id (and the triangles in post-build order if asked)."""
        tris = np.ascontiguousarray(tris, dtype=np.float32)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert tris.ndim == 2 and tris.shape[1] == TRI_FLOATS and rgba.ndim == 3 and rgba.shape[2] == 4
        out = np.empty_like(tris) if want_triangles else None
        mid = C.c_int32(-1)
        self._check(lib().rt_mesh_upload_soup(self._h, tris.ctypes.data, int(tris.shape[0]), rgba.ctypes.data, int(rgba.shape[1]),
                                              int(rgba.shape[0]), C.byref(mid), out.ctypes.data if want_triangles else None))
        return (mid.value, out) if want_triangles else mid.value

    def rebuild_mesh(self, mesh_id: int, tris: np.ndarray, want_triangles: bool = False):
        """rt_mesh_rebuild: new triangles (file order) for an uploaded mesh, BVH rebuilt; the texture stays."""
        tris = np.ascontiguousarray(tris, dtype=np.float32)
        assert tris.ndim == 2 and tris.shape[1] == TRI_FLOATS
        out = np.empty_like(tris) if want_triangles else None
        self._check(lib().rt_mesh_rebuild(self._h, int(mesh_id), tris.ctypes.data, int(tris.shape[0]), out.ctypes.data if want_triangles else None))
        return out

    def upload_host_mesh(self, mesh) -> int:
        """Uploads a Mesh struct of raytracer_amd.host_api (what create_mesh returns) as it stands."""
        mid = C.c_int32(-1)
        self._check(lib().rt_mesh_upload_tree(self._h, C.cast(mesh.triangles, C.c_void_p), int(mesh.triangle_count),
                                              C.cast(mesh.bvh.root, C.c_void_p), C.cast(mesh.texture_data, C.c_void_p),
                                              int(mesh.texture_width), int(mesh.texture_height), C.byref(mid)))
        return mid.value

    def clear_scene(self):
        self._check(lib().rt_scene_clear(self._h))

    def mesh_count(self) -> int:
        return lib().rt_mesh_count(self._h)

    def set_option(self, name: str, value: int):
        self._check(lib().rt_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self._check(lib().rt_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def render(self, camera: Camera, light: Light, transforms: Sequence[Transform], width: int, height: int,
               rgb, hit_id=None, hit_t=None, shadow=None, flags: int = 0, stream: Optional[int] = None):
        """Asynchronous launch into device buffers (torch tensors or raw device pointers)."""
        arr = (Transform * len(transforms))(*transforms)
        self._check(lib().rt_render_frame(self._h, C.byref(camera), C.byref(light), arr, len(transforms), int(width),
                                          int(height), int(flags), _dev_ptr(rgb), _dev_ptr(hit_id), _dev_ptr(hit_t),
                                          _dev_ptr(shadow), stream))

    def render_frames(self, camera: Camera, light: Light, transforms: Sequence[Sequence[Transform]], width: int, height: int,
                      rgb, frame_stride: Optional[int] = None, stream: Optional[int] = None, noalloc: bool = False):
        """rt_render_frames: transforms[f] are the placements of frame f; frame f lands at rgb + f*frame_stride
        (default: images back to back, i.e. a [n_frames, height, width, 3] uint8 tensor).
        noalloc=True calls rt_render_frames_noalloc (same result, no device allocation inside the call)."""
        n = len(transforms)
        m = len(transforms[0]) if n else 0
        flat = [t for frame in transforms for t in frame]
        if len(flat) != n * m:
            raise ValueError("every frame needs the same number of transforms")
        arr = (Transform * max(len(flat), 1))(*flat)
        stride = int(frame_stride) if frame_stride is not None else int(width) * int(height) * 3
        entry = lib().rt_render_frames_noalloc if noalloc else lib().rt_render_frames
        self._check(entry(self._h, C.byref(camera), C.byref(light), arr, n, m, int(width), int(height), _dev_ptr(rgb), stride, stream))

    def render_host(self, camera: Camera, light: Light, transforms: Sequence[Transform], width: int, height: int,
                    debug: bool = True, flags: int = 0):
        """Synchronous render into fresh numpy arrays: (rgb[h,w,3], hit_id, hit_t, shadow)."""
        arr = (Transform * len(transforms))(*transforms)
        rgb = np.zeros((height, width, 3), np.uint8)
        hid = np.zeros((height, width), np.int32) if debug else None
        ht = np.zeros((height, width), np.float32) if debug else None
        sh = np.zeros((height, width), np.uint8) if debug else None
        self._check(lib().rt_render_frame_host(self._h, C.byref(camera), C.byref(light), arr, len(transforms), int(width),
                                               int(height), int(flags), rgb.ctypes.data,
                                               hid.ctypes.data if debug else None, ht.ctypes.data if debug else None,
                                               sh.ctypes.data if debug else None))
        return rgb, hid, ht, sh

    def render_sampled(self, camera: Camera, lights: Sequence[Light], transforms: Sequence[Transform], width: int, height: int,
                       samples_per_pixel: int, seed: int, rgb, hit_id=None, flags: int = 0, stream: Optional[int] = None):
        """rt_render_frame_sampled into device buffers (asynchronous)."""
        arr = (Transform * max(len(transforms), 1))(*transforms)
        larr = (Light * max(len(lights), 1))(*lights)
        self._check(lib().rt_render_frame_sampled(self._h, C.byref(camera), larr, len(lights), arr, len(transforms), int(width),
                                                  int(height), int(samples_per_pixel), int(seed) & 0xFFFFFFFF, int(flags),
                                                  _dev_ptr(rgb), _dev_ptr(hit_id), stream))

    def render_strips(self, camera: Camera, light: Light, transforms: Sequence[Transform], width: int, height: int,
                      strip_stride: int, strip_offset: int, rgb, hit_id=None, hit_t=None, shadow=None, flags: int = 0,
                      stream: Optional[int] = None):
        """rt_render_frame_strips: this share of the frame's eight-row strips into planes strip_rows(...) rows tall."""
        arr = (Transform * max(len(transforms), 1))(*transforms)
        self._check(lib().rt_render_frame_strips(self._h, C.byref(camera), C.byref(light), arr, len(transforms), int(width), int(height),
                                                 int(strip_stride), int(strip_offset), int(flags), _dev_ptr(rgb), _dev_ptr(hit_id),
                                                 _dev_ptr(hit_t), _dev_ptr(shadow), stream))

    def render_sampled_strips(self, camera: Camera, lights: Sequence[Light], transforms: Sequence[Transform], width: int, height: int,
                              samples_per_pixel: int, seed: int, strip_stride: int, strip_offset: int, rgb, hit_id=None, flags: int = 0,
                              stream: Optional[int] = None):
        """rt_render_frame_sampled_strips (see render_strips)."""
        arr = (Transform * max(len(transforms), 1))(*transforms)
        larr = (Light * max(len(lights), 1))(*lights)
        self._check(lib().rt_render_frame_sampled_strips(self._h, C.byref(camera), larr, len(lights), arr, len(transforms), int(width),
                                                         int(height), int(samples_per_pixel), int(seed) & 0xFFFFFFFF, int(strip_stride),
                                                         int(strip_offset), int(flags), _dev_ptr(rgb), _dev_ptr(hit_id), stream))

    @staticmethod
    def make_instances(mesh_ids: Sequence[int], transforms: Sequence[Transform]):
        """ctypes array of rt_instance for render_instanced (build once, reuse across calls)."""
        arr = (Instance * max(len(mesh_ids), 1))()
        for i, (m, t) in enumerate(zip(mesh_ids, transforms)):
            arr[i].mesh = int(m)
            arr[i].transform = t
        return arr, len(mesh_ids)

    def render_instanced(self, camera: Camera, light: Light, instances, width: int, height: int, rgb, hit_instance=None,
                         hit_tri=None, hit_t=None, shadow=None, flags: int = 0, stream: Optional[int] = None):
        """rt_render_frame_instanced; `instances` is what make_instances returns."""
        arr, n = instances
        self._check(lib().rt_render_frame_instanced(self._h, C.byref(camera), C.byref(light), arr, n, int(width), int(height),
                                                    int(flags), _dev_ptr(rgb), _dev_ptr(hit_instance), _dev_ptr(hit_tri),
                                                    _dev_ptr(hit_t), _dev_ptr(shadow), stream))

    def upscale_frames(self, rgb, n_frames: int, width: int, height: int, scaled_width: int, scaled_height: int, argb,
                       frame_stride: Optional[int] = None, out_stride: Optional[int] = None, stream: Optional[int] = None):
        """rt_upscale_frames on device buffers: rgb [n, h, w, 3] uint8 -> argb [n, sh, sw] 32-bit."""
        fs = int(frame_stride) if frame_stride is not None else int(width) * int(height) * 3
        os_ = int(out_stride) if out_stride is not None else int(scaled_width) * int(scaled_height)
        self._check(lib().rt_upscale_frames(self._h, _dev_ptr(rgb), fs, int(n_frames), int(width), int(height),
                                            int(scaled_width), int(scaled_height), _dev_ptr(argb), os_, stream))

    def upscale_host(self, rgb: np.ndarray, scaled_width: int, scaled_height: int) -> np.ndarray:
        """rt_upscale_frame_host: numpy [h, w, 3] uint8 -> numpy [sh, sw] uint32 (ARGB)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        out = np.zeros((int(scaled_height), int(scaled_width)), np.uint32)
        self._check(lib().rt_upscale_frame_host(self._h, rgb.ctypes.data, int(rgb.shape[1]), int(rgb.shape[0]),
                                                int(scaled_width), int(scaled_height), out.ctypes.data))
        return out

    def set_trace_buffer(self, buf):
        """Per-tile {start, end} clock trace into a device buffer of 2 uint64 per 8x8 tile (None = off)."""
        self._check(lib().rt_set_trace_buffer(self._h, _dev_ptr(buf)))

    def sync(self, stream: Optional[int] = None):
        self._check(lib().rt_sync(self._h, stream))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        self._check(lib().rt_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def build_bvh(self, tris: np.ndarray):
        """rt_bvh_build_device on a copy: (triangles in post-build order [n,24] f32, nodes NODE_DTYPE), built on the GPU."""
        t = np.ascontiguousarray(tris, dtype=np.float32).copy()
        n = int(t.shape[0])
        nodes = np.zeros(max(2 * n, 1), dtype=NODE_DTYPE)
        used = C.c_int32(0)
        self._check(lib().rt_bvh_build_device(self._h, t.ctypes.data, n, nodes.ctypes.data, int(nodes.shape[0]), C.byref(used)))
        return t, nodes[:used.value].copy()

    def build_stats(self) -> dict:
        ms, levels, launches = C.c_float(0), C.c_int32(0), C.c_int32(0)
        self._check(lib().rt_bvh_build_stats(self._h, C.byref(ms), C.byref(levels), C.byref(launches)))
        return {"device_ms": ms.value, "levels": levels.value, "launches": launches.value}

    def last_counters(self) -> dict:
        c = Counters()
        self._check(lib().rt_last_counters(self._h, C.byref(c)))
        return c.as_dict()
