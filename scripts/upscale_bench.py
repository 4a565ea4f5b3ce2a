#!/usr/bin/env python3
"""Times the bicubic upscale kernel (rt_upscale_frames) on frames that are resident in HBM and
prints one JSON line with its roofline and the CPU baseline (the compiled reference's
bicubic_interpolate in the reference's loop, or the CPU restatement where it is absent).

    python scripts/upscale_bench.py --frames 16 --scale 0.9
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0
VALU_PEAK_TFLOPS = 157.3
FLOP_PER_PIXEL = 15 * 27 + 6        # 15 cubics of 27 operations (image.c:4-11) + rounding adds, per output pixel


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--scale", type=float, default=0.9)
    p.add_argument("--frames", type=int, default=16, help="frames per launch")
    p.add_argument("--reps", type=int, default=20)
    p.add_argument("--no-cpu-baseline", action="store_true")
    a = p.parse_args()
    from raytracer_amd import capi, demo
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = demo.DemoScene(0)
    ctx = scene.ctx
    W, H = a.width, a.height
    SW, SH = capi.scaled_size(W, H, a.scale)
    stream = torch.cuda.Stream(device=dev)
    rgb = torch.empty((a.frames, H, W, 3), dtype=torch.uint8, device=dev)
    argb = torch.empty((a.frames, SH, SW), dtype=torch.int32, device=dev)
    for i in range(0, a.frames, 8):
        fr = list(range(i, min(i + 8, a.frames)))
        scene.render_frames(fr, W, H, rgb[i:i + len(fr)], stream=stream.cuda_stream)
    stream.synchronize()
    ms = []
    for r in range(a.reps + 2):
        ctx.upscale_frames(rgb, a.frames, W, H, SW, SH, argb, stream=stream.cuda_stream)
        t = ctx.last_kernel_ms()
        if r >= 2:
            ms.append(t)
    kern_ms = float(np.mean(ms))
    # wall clock of back-to-back launches
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ctx.set_option("timing", 0)
    for r in range(a.reps):
        ctx.upscale_frames(rgb, a.frames, W, H, SW, SH, argb, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    wall_ms = (time.perf_counter() - t0) * 1e3 / a.reps
    alg_bytes = a.frames * (W * H * 3 + SW * SH * 4)
    gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
    tfl = a.frames * SW * SH * FLOP_PER_PIXEL / (kern_ms * 1e-3) / 1e12
    out = {
        "metric": "bicubic upscale, Mpixels/s out", "value": round(a.frames * SW * SH / (wall_ms * 1e-3) / 1e6, 1), "unit": "Mpixel/s",
        "config": {"workload": f"{a.frames} frames {W}x{H} RGB8 -> {SW}x{SH} ARGB32 per launch (scale_factor {a.scale})"},
        "ms_per_launch": round(wall_ms, 4), "ms_per_frame": round(wall_ms / a.frames, 5), "dtype": "f32",
        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": "upscale_frames<8>", "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "valu": {"achieved": round(tfl, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / VALU_PEAK_TFLOPS, 4),
                              "basis": "reference-equivalent operations: 15 cubics per output pixel"}},
    }
    if not a.no_cpu_baseline:
        from oracle import pyoracle as po
        frame = rgb[0].cpu().numpy()
        use_ref = po.ref_available()
        threads = po.ref().refc_max_threads() if use_ref else po.lib().orc_max_threads()
        po.upscale_frame(frame, SW, SH, use_ref=use_ref)
        n = 4
        t0 = time.perf_counter()
        for _ in range(n):
            po.upscale_frame(frame, SW, SH, use_ref=use_ref)
        dt = (time.perf_counter() - t0) / n
        out["cpu_baseline"] = {"value": round(SW * SH / dt / 1e6, 1), "unit": "Mpixel/s", "cores": int(threads),
                               "kind": "reference" if use_ref else "port", "ms_per_frame": round(dt * 1e3, 2),
                               "sample": f"{n} frames {W}x{H} -> {SW}x{SH}, OpenMP static rows as scene.c:188"}
    print(json.dumps(out), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
