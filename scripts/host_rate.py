#!/usr/bin/env python3
"""Frame rate with the framebuffer delivered to HOST memory (the PCIe-inclusive figure):
rt_render_frame_host per frame, and the reference host API's render_scene() loop.

    python scripts/host_rate.py --frames 48
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402,F401


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--frames", type=int, default=48)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--in-flight", type=lambda v: [int(x) for x in v.split(",") if x], default=[2, 4, 8])
    a = p.parse_args()
    from raytracer_amd import demo, host_api as H
    W, Hh = a.width, a.height
    scene = demo.DemoScene(0)
    tfs = [scene.transforms(f % demo.N_FRAMES) for f in range(a.frames)]
    scene.ctx.render_host(scene.camera, scene.light, tfs[0], W, Hh, debug=False)
    t0 = time.perf_counter()
    hits = 0
    for t in tfs:
        scene.ctx.render_host(scene.camera, scene.light, t, W, Hh, debug=False)
    dt = time.perf_counter() - t0
    out = {"rt_render_frame_host_ms_per_frame": round(dt / a.frames * 1e3, 3)}
    scene.close()

    L = H.lib()
    sc = L.create_scene(W, Hh, a.frames * 1000 // 24, 24, 1.0)
    L.set_scene_camera(C.byref(sc), H.Vec3(-3, 3, -3), H.Vec3(0, 0, 0), H.Vec3(0, 1, 0), 60.0)
    L.set_scene_light(C.byref(sc), H.Vec3(1, 1, -1), H.Vec3(1.4, 1.4, 1.4))
    loaded = []
    for n in demo.MESH_NAMES:
        m = L.create_mesh(os.path.join(H.ASSETS, n + ".obj").encode(), os.path.join(H.ASSETS, n + ".webp").encode())
        loaded.append(m)
        L.add_mesh_to_scene(C.byref(sc), m)
    L.render_scene(C.byref(sc))                      # first call uploads the scene

    def animation():                                 # the driver's loop, raytracer.c:36-59
        sc.current_frame = 0
        t0 = time.perf_counter()
        for f in range(sc.frame_count):
            d, t = H.Transform(), H.Transform()
            L.rt_host_demo_animation(f, C.byref(d), C.byref(t))
            L.set_mesh_position(C.byref(sc.meshes[0]), d.position)
            L.set_mesh_rotation(C.byref(sc.meshes[0]), d.rotation)
            L.set_mesh_position(C.byref(sc.meshes[1]), t.position)
            L.set_mesh_rotation(C.byref(sc.meshes[1]), t.rotation)
            L.render_scene(C.byref(sc))
            L.next_frame(C.byref(sc))
        assert L.rt_host_wait(C.byref(sc)) == 0
        dt = time.perf_counter() - t0
        frames = [bytes(C.string_at(sc.frames[f], W * Hh * 3)) for f in range(sc.frame_count)]
        return dt, frames

    dt, want = animation()
    out["host_api_render_scene_ms_per_frame"] = round(dt / sc.frame_count * 1e3, 3)
    for n in a.in_flight:
        L.rt_host_set_frames_in_flight(n)
        animation()                                  # the ring's streams and images are made on first use
        dt, got = animation()
        out[f"host_api_render_scene_ms_per_frame_{n}_in_flight"] = round(dt / sc.frame_count * 1e3, 3)
        out[f"same_frames_{n}_in_flight"] = got == want
    L.rt_host_set_frames_in_flight(1)
    out["frames"] = int(sc.frame_count)
    out["size"] = [W, Hh]
    for m in loaded:
        L.destroy_mesh(C.byref(m))
    L.destroy_scene(C.byref(sc))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
