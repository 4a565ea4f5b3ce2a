#!/usr/bin/env python3
"""Renders a few 1080p frames with one kernel variant and nothing else: the program that
rocprofv3 (kernel trace or --pmc passes) is pointed at.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 scripts/prof_frames.py --variant 0
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def other_modes(a, scene, dev):
    import numpy as np
    from raytracer_amd import capi, demo, host_api as H
    ctx = scene.ctx
    frames = [int(f) for f in a.frames.split(",")]
    if a.mode == "sampled":
        L = H.lib()
        lights = []
        for d, c in (((1, 1, -1), (1.4, 1.4, 1.4)), ((-1, 2, 0.5), (0.3, 0.5, 0.9))):
            l = L.create_directional_light(H.Vec3(*d), H.Vec3(*c))
            lights.append(capi.Light(capi.Vec3(l.direction.x, l.direction.y, l.direction.z), capi.Vec3(l.color.x, l.color.y, l.color.z)))
        rgb = torch.empty((2160, 3840, 3), dtype=torch.uint8, device=dev)
        for f in frames[:2]:
            ctx.render_sampled(scene.camera, lights, scene.transforms(f), 3840, 2160, 16, 0, rgb)
    elif a.mode == "instanced":
        rng = np.random.default_rng(3)
        ids, tfs = [0, 2], [None, capi.Transform()]
        for gx in range(56):
            for gz in range(56):
                ids.append(1)
                tfs.append(capi.Transform(capi.Vec3((gx - 27.5) * 1.05, 0.45 + 0.1 * rng.uniform(), (gz - 27.5) * 1.05),
                                          capi.Vec3(0.1 * rng.uniform(), rng.uniform(0, 6.28), 0)))
        rgb = torch.empty((a.height, a.width, 3), dtype=torch.uint8, device=dev)
        for f in frames:
            tfs[0] = scene.transforms(f)[0]
            ctx.render_instanced(scene.camera, scene.light, capi.Context.make_instances(ids, tfs), a.width, a.height, rgb)
    else:
        sw, sh = capi.scaled_size(a.width, a.height, 0.9)
        rgb = torch.empty((16, a.height, a.width, 3), dtype=torch.uint8, device=dev)
        argb = torch.empty((16, sh, sw), dtype=torch.int32, device=dev)
        scene.render_frames(list(range(8)), a.width, a.height, rgb[:8])
        scene.render_frames(list(range(8, 16)), a.width, a.height, rgb[8:])
        for _ in range(4):
            ctx.upscale_frames(rgb, 16, a.width, a.height, sw, sh, argb)
    ctx.sync()
    torch.cuda.synchronize()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--variant", type=int, default=-1)
    p.add_argument("--block", type=int, default=0)
    p.add_argument("--frames", default="0,45,90")
    p.add_argument("--repeat", type=int, default=2)
    p.add_argument("--batch", type=int, default=0, help="> 0: rt_render_frames with this many consecutive frames per launch")
    p.add_argument("--mode", default="frames", choices=["frames", "bench", "sampled", "instanced", "upscale"],
                   help="bench: the launches of bench.py's timed region (96 steps, bench.FRAMES_PER_LAUNCH frames per launch); sampled: 4K/16 samples/2 lights; instanced: 56x56 treasures; upscale: 16 frames 1080p -> scale 0.9")
    p.add_argument("--wave", type=int, default=-1, help="option instanced_wave (instanced mode)")
    p.add_argument("--count-packed", action="store_true",
                   help="bench mode: no launches for the counters; prints the two-wide packed float operations per launch as the kernel's own "
                        "work counters see them (six per box test), the fifth argument of scripts/pmc_to_profile.py")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    a = p.parse_args()
    from raytracer_amd import demo
    dev = torch.device("cuda", 0)
    scene = demo.DemoScene(0)
    if a.variant >= 0:
        scene.ctx.set_option("variant", a.variant)
    if a.block:
        scene.ctx.set_option("block", a.block)
    scene.ctx.set_option("timing", 0)
    if a.wave >= 0:
        scene.ctx.set_option("instanced_wave", a.wave)
    if a.mode == "bench":
        import bench
        seq = bench.frame_sequence(list(range(demo.N_FRAMES)), 96)
        n = bench.FRAMES_PER_LAUNCH
        buf = torch.empty((n, a.height, a.width, 3), dtype=torch.uint8, device=dev)
        if a.count_packed:
            from raytracer_amd import capi
            boxes = {}
            for f in sorted(set(seq)):
                scene.render(f, a.width, a.height, buf[0], flags=capi.RT_FLAG_COUNT_WORK)
                scene.ctx.sync()
                boxes[f] = scene.ctx.last_counters()["slab_tests"]
            per_launch = [sum(boxes[f] for f in seq[b:b + n]) for b in range(0, 96, n)]
            print(int(6 * sum(per_launch) / len(per_launch)))
            scene.close()
            return
        for _ in range(a.repeat):
            for b in range(0, 96, n):
                scene.render_frames(seq[b:b + n], a.width, a.height, buf)
        scene.ctx.sync()
        torch.cuda.synchronize()
        scene.close()
        return
    if a.mode != "frames":
        other_modes(a, scene, dev)
        scene.close()
        return
    buf = torch.empty((max(a.batch, 1), a.height, a.width, 3), dtype=torch.uint8, device=dev)
    for _ in range(a.repeat):
        for f in a.frames.split(","):
            if a.batch > 0:
                scene.render_frames([(int(f) + i) % demo.N_FRAMES for i in range(a.batch)], a.width, a.height, buf)
            else:
                scene.render(int(f), a.width, a.height, buf[0])
    scene.ctx.sync()
    torch.cuda.synchronize()
    scene.close()


if __name__ == "__main__":
    main()
