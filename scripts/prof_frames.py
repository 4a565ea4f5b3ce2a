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


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--variant", type=int, default=-1)
    p.add_argument("--block", type=int, default=0)
    p.add_argument("--frames", default="0,45,90")
    p.add_argument("--repeat", type=int, default=2)
    p.add_argument("--batch", type=int, default=0, help="> 0: rt_render_frames with this many consecutive frames per launch")
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
