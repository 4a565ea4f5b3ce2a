"""The reference driver's scene (raytracer.c:10-53) on the product libraries: drone, treasure and
ground meshes loaded with the host API's create_mesh, camera and light as the driver sets them,
and the per-frame placements of its animation loop."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import sys
from typing import List

from . import capi, host_api

MESH_NAMES = ("drone", "treasure", "ground")          # raytracer.c:27-34, in add_mesh_to_scene order
N_FRAMES = (4000 * 24) // 1000                        # raytracer.c:10 -> scene.c:9


@contextlib.contextmanager
def stdout_to_stderr():
    """create_mesh reports what it loaded with printf, like the reference (mesh.c:112).  Programs that keep
    stdout for one JSON line send those lines to stderr while the scene loads."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class DemoScene:
    def __init__(self, device: int = 0, quiet_stdout: bool = True):
        H = host_api.lib()
        self.ctx = capi.Context(device)
        self.meshes = []
        with (stdout_to_stderr() if quiet_stdout else contextlib.nullcontext()):
            for name in MESH_NAMES:
                m = H.create_mesh(os.path.join(host_api.ASSETS, name + ".obj").encode(),
                                  os.path.join(host_api.ASSETS, name + ".webp").encode())
                if not m.bvh.root or not m.texture_data:
                    raise RuntimeError(f"could not load {name}")
                self.meshes.append(m)
                self.ctx.upload_host_mesh(m)
        cam = H.create_camera(host_api.Vec3(-3, 3, -3), host_api.Vec3(0, 0, 0), host_api.Vec3(0, 1, 0), 60.0)   # raytracer.c:13-18
        lit = H.create_directional_light(host_api.Vec3(1, 1, -1), host_api.Vec3(1.4, 1.4, 1.4))                 # raytracer.c:21-24
        self.camera = capi.Camera(capi.Vec3(cam.position.x, cam.position.y, cam.position.z),
                                  capi.Vec3(cam.look_at.x, cam.look_at.y, cam.look_at.z),
                                  capi.Vec3(cam.up.x, cam.up.y, cam.up.z), cam.fov)
        self.light = capi.Light(capi.Vec3(lit.direction.x, lit.direction.y, lit.direction.z),
                                capi.Vec3(lit.color.x, lit.color.y, lit.color.z))

    def transforms(self, frame: int) -> List[capi.Transform]:
        drone, treasure = host_api.Transform(), host_api.Transform()
        host_api.lib().rt_host_demo_animation(int(frame), C.byref(drone), C.byref(treasure))
        out = []
        for t in (drone, treasure, host_api.Transform()):
            out.append(capi.Transform(capi.Vec3(t.position.x, t.position.y, t.position.z),
                                      capi.Vec3(t.rotation.x, t.rotation.y, t.rotation.z)))
        return out

    def render(self, frame: int, width: int, height: int, rgb, hit_id=None, hit_t=None, shadow=None,
               flags: int = 0, stream=None):
        self.ctx.render(self.camera, self.light, self.transforms(frame), width, height, rgb, hit_id, hit_t, shadow,
                        flags=flags, stream=stream)

    def render_frames(self, frames, width: int, height: int, rgb, stream=None):
        """One launch for several animation frames; rgb is a [len(frames), height, width, 3] uint8 device tensor."""
        self.ctx.render_frames(self.camera, self.light, [self.transforms(f) for f in frames], width, height, rgb,
                               stream=stream)

    def close(self):
        H = host_api.lib()
        self.ctx.close()
        for m in self.meshes:
            H.destroy_mesh(C.byref(m))
        self.meshes = []
