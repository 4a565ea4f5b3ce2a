"""ctypes binding of libraytracer_host.so: the reference ray tracer's own host API
(create_scene, create_mesh, add_mesh_to_scene, set_mesh_position, render_scene ...), whose
render_scene() runs on the GPU through librt_mi355.so.

The function names, argument order and struct layouts are the reference's
(scene.h:29-36, geometry/mesh.h:22-26, accel/bvh.h:23-28, render/camera.h:15, render/light.h:12),
so a driver reads like the reference's main() (raytracer.c:8-75).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libraytracer_host.so")
ASSETS = os.path.join(_HERE, "assets")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Vec2(C.Structure):
    _fields_ = [("u", C.c_float), ("v", C.c_float)]


class Transform(C.Structure):
    _fields_ = [("position", Vec3), ("rotation", Vec3)]


class AABB(C.Structure):
    _fields_ = [("min", Vec3), ("max", Vec3)]


class Triangle(C.Structure):
    _fields_ = [("v0", Vec3), ("v1", Vec3), ("v2", Vec3), ("t0", Vec2), ("t1", Vec2), ("t2", Vec2),
                ("n0", Vec3), ("n1", Vec3), ("n2", Vec3)]


class BVHNode(C.Structure):
    pass


BVHNode._fields_ = [("bounds", AABB), ("left", C.POINTER(BVHNode)), ("right", C.POINTER(BVHNode)),
                    ("start_idx", C.c_int), ("triangle_count", C.c_int)]


class BVH(C.Structure):
    _fields_ = [("root", C.POINTER(BVHNode)), ("triangles", C.POINTER(Triangle)), ("triangle_count", C.c_size_t)]


class Mesh(C.Structure):
    _fields_ = [("triangles", C.POINTER(Triangle)), ("triangle_count", C.c_size_t),
                ("texture_data", C.POINTER(C.c_ubyte)), ("texture_width", C.c_int), ("texture_height", C.c_int),
                ("bvh", BVH), ("transform", Transform)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("look_at", Vec3), ("up", Vec3), ("fov", C.c_float)]


class DirectionalLight(C.Structure):
    _fields_ = [("direction", Vec3), ("color", Vec3)]


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class Mat4(C.Structure):
    _fields_ = [("m", (C.c_float * 4) * 4)]


class Scene(C.Structure):
    _fields_ = [("meshes", C.POINTER(Mesh)), ("mesh_count", C.c_size_t), ("camera", Camera), ("light", DirectionalLight),
                ("frames", C.POINTER(C.POINTER(C.c_ubyte))), ("frame_count", C.c_int), ("current_frame", C.c_int),
                ("width", C.c_int), ("height", C.c_int), ("scale_factor", C.c_float), ("duration_ms", C.c_int),
                ("fps", C.c_int)]


assert C.sizeof(Triangle) == 96 and C.sizeof(BVHNode) == 48 and C.sizeof(Camera) == 40

EXPORTS = [
    "create_scene", "add_mesh_to_scene", "set_scene_camera", "set_scene_light", "next_frame", "render_scene",
    "save_scene", "destroy_scene", "create_mesh", "set_mesh_position", "set_mesh_rotation", "destroy_mesh",
    "sample_mesh_texture", "create_bvh_node", "create_bvh", "destroy_bvh_node", "destroy_bvh", "intersect_bvh", "create_camera",
    "create_directional_light", "rt_host_set_device", "rt_host_last_status", "rt_host_render_debug",
    "rt_host_demo_animation", "rt_host_write_ppm", "rt_host_scaled_size", "rt_host_upscale_frame",
    "rt_host_set_obj_mode", "rt_host_invalidate_scene", "rt_host_set_gpus", "rt_host_animation_has_webp", "rt_host_animation_open", "rt_host_animation_add",
    "rt_host_animation_close", "rt_host_set_save_options", "rt_host_set_frames_in_flight", "rt_host_wait",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make product`")
        from .capi import _load_torch_hip_first
        _load_torch_hip_first()
        L = C.CDLL(LIB_PATH)
        L.create_scene.restype = Scene
        L.create_scene.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        L.add_mesh_to_scene.restype = None
        L.add_mesh_to_scene.argtypes = [C.POINTER(Scene), Mesh]
        L.set_scene_camera.restype = None
        L.set_scene_camera.argtypes = [C.POINTER(Scene), Vec3, Vec3, Vec3, C.c_float]
        L.set_scene_light.restype = None
        L.set_scene_light.argtypes = [C.POINTER(Scene), Vec3, Vec3]
        for name in ("next_frame", "render_scene", "destroy_scene"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.POINTER(Scene)]
        L.save_scene.restype = None
        L.save_scene.argtypes = [C.POINTER(Scene), C.c_char_p]
        L.create_mesh.restype = Mesh
        L.create_mesh.argtypes = [C.c_char_p, C.c_char_p]
        L.set_mesh_position.restype = None
        L.set_mesh_position.argtypes = [C.POINTER(Mesh), Vec3]
        L.set_mesh_rotation.restype = None
        L.set_mesh_rotation.argtypes = [C.POINTER(Mesh), Vec3]
        L.destroy_mesh.restype = None
        L.destroy_mesh.argtypes = [C.POINTER(Mesh)]
        L.sample_mesh_texture.restype = Vec3
        L.sample_mesh_texture.argtypes = [C.POINTER(Mesh), C.c_float, C.c_float]
        L.create_bvh_node.restype = C.POINTER(BVHNode)
        L.create_bvh_node.argtypes = [C.POINTER(Triangle), C.c_int, C.c_int]
        L.create_bvh.restype = BVH
        L.create_bvh.argtypes = [C.POINTER(Triangle), C.c_size_t]
        L.destroy_bvh_node.restype = None
        L.destroy_bvh_node.argtypes = [C.POINTER(BVHNode)]
        L.destroy_bvh.restype = None
        L.destroy_bvh.argtypes = [C.POINTER(BVH)]
        L.intersect_bvh.restype = C.c_bool
        L.intersect_bvh.argtypes = [C.POINTER(BVHNode), Ray, C.POINTER(Triangle)] + [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_int)]
        L.create_camera.restype = Camera
        L.create_camera.argtypes = [Vec3, Vec3, Vec3, C.c_float]
        L.create_directional_light.restype = DirectionalLight
        L.create_directional_light.argtypes = [Vec3, Vec3]
        L.rt_host_set_device.restype = None
        L.rt_host_set_device.argtypes = [C.c_int]
        L.rt_host_last_status.argtypes = [C.POINTER(Scene)]
        L.rt_host_render_debug.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_host_demo_animation.restype = None
        L.rt_host_demo_animation.argtypes = [C.c_int, C.POINTER(Transform), C.POINTER(Transform)]
        L.rt_host_write_ppm.argtypes = [C.POINTER(Scene), C.c_int, C.c_char_p]
        L.rt_host_animation_has_webp.restype = C.c_int
        L.rt_host_animation_has_webp.argtypes = []
        L.rt_host_animation_open.restype = C.c_int
        L.rt_host_animation_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_host_animation_add.restype = C.c_int
        L.rt_host_animation_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.rt_host_animation_close.restype = C.c_int
        L.rt_host_animation_close.argtypes = [C.c_void_p]
        L.rt_host_set_save_options.restype = None
        L.rt_host_set_save_options.argtypes = [C.c_float, C.c_int]
        L.rt_host_set_frames_in_flight.restype = None
        L.rt_host_set_frames_in_flight.argtypes = [C.c_int]
        L.rt_host_wait.restype = C.c_int
        L.rt_host_wait.argtypes = [C.POINTER(Scene)]
        L.rt_host_set_gpus.restype = C.c_int
        L.rt_host_set_gpus.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.rt_host_invalidate_scene.restype = None
        L.rt_host_invalidate_scene.argtypes = [C.POINTER(Scene)]
        L.rt_host_set_obj_mode.restype = None
        L.rt_host_set_obj_mode.argtypes = [C.c_int]
        L.rt_host_scaled_size.argtypes = [C.POINTER(Scene), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rt_host_upscale_frame.argtypes = [C.POINTER(Scene), C.c_int, C.c_void_p]
        # value-type helpers of the reference's math / aabb / camera / image headers
        v3, f = Vec3, C.c_float
        for name, res, args in (
                ("vec3_add", v3, [v3, v3]), ("vec3_sub", v3, [v3, v3]), ("vec3_mul", v3, [v3, f]), ("vec3_div", v3, [v3, f]),
                ("vec3_dot", f, [v3, v3]), ("vec3_cross", v3, [v3, v3]), ("vec3_length", f, [v3]), ("vec3_normalize", v3, [v3]),
                ("vec3_mul_vec3", v3, [v3, v3]), ("mat4_identity", Mat4, []), ("mat4_multiply", Mat4, [Mat4, Mat4]),
                ("mat4_translation", Mat4, [v3]), ("mat4_rotation_x", Mat4, [f]), ("mat4_rotation_y", Mat4, [f]),
                ("mat4_rotation_z", Mat4, [f]), ("mat4_transform_point", v3, [Mat4, v3]), ("mat4_transform_vector", v3, [Mat4, v3]),
                ("mat4_inverse", Mat4, [Mat4]), ("mat4_transpose", Mat4, [Mat4]), ("transform_ray", Ray, [Ray, Transform]),
                ("transform_normal", v3, [v3, Transform]),
                ("ray_triangle_intersect", C.c_bool, [Ray, v3, v3, v3, C.POINTER(f), C.POINTER(f), C.POINTER(f)]),
                ("create_empty_aabb", AABB, []), ("expand_aabb", AABB, [AABB, v3]), ("get_triangle_bounds", AABB, [Triangle]),
                ("ray_aabb_intersect", C.c_bool, [Ray, AABB]), ("get_camera_ray", Ray, [C.POINTER(Camera), f, f, f]),
                ("cubic_hermite", f, [f, f, f, f, f]), ("get_pixel_rgb", C.c_uint32, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
                ("bicubic_interpolate", C.c_uint32, [C.c_void_p, f, f, C.c_int, C.c_int])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        for name in EXPORTS:
            getattr(L, name)
        _lib = L
    return _lib


def frame_array(scene: Scene, index: int) -> np.ndarray:
    """Copy of scene.frames[index] as [height, width, 3] uint8."""
    n = scene.width * scene.height * 3
    return np.ctypeslib.as_array(scene.frames[index], shape=(n,)).reshape(scene.height, scene.width, 3).copy()


def mesh_triangles(mesh: Mesh) -> np.ndarray:
    """Copy of the mesh's triangle array as [n, 24] float32 (post-build order)."""
    n = int(mesh.triangle_count)
    if n == 0:
        return np.zeros((0, 24), np.float32)
    return np.ctypeslib.as_array(C.cast(mesh.triangles, C.POINTER(C.c_float)), shape=(n, 24)).copy()


def mesh_texture(mesh: Mesh) -> np.ndarray:
    return np.ctypeslib.as_array(mesh.texture_data, shape=(mesh.texture_height, mesh.texture_width, 4)).copy()
