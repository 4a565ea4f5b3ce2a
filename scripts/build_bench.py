#!/usr/bin/env python3
"""Times rt_bvh_build_device against rt_bvh_build (host) on random triangle soups and checks that both
leave the same bits.  One JSON line.

    python scripts/build_bench.py --sizes 3155,100000,1000000,10000000
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401


def soup_of(n, seed):
    rng = np.random.default_rng(seed)
    t = np.zeros((n, 24), np.float32)
    c = rng.random((n, 1, 3), dtype=np.float32) * 100
    t[:, :9] = (c + rng.random((n, 3, 3), dtype=np.float32) * 0.1).reshape(n, 9)
    t[:, 9:] = rng.random((n, 15), dtype=np.float32)
    return t


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--sizes", default="3155,100000,1000000,10000000")
    p.add_argument("--reps", type=int, default=3)
    p.add_argument("--no-host", action="store_true", help="skip the host build (and the comparison)")
    a = p.parse_args()
    from raytracer_amd import capi
    ctx = capi.Context(0)
    rows = []
    for n in [int(x) for x in a.sizes.split(",")]:
        soup = soup_of(n, n % 1000)
        ctx.build_bvh(soup[: min(n, 1000)])                       # warm-up: module load, first allocations
        dev_wall, dev_kernels = [], []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            got_t, got_n = ctx.build_bvh(soup)
            dev_wall.append(time.perf_counter() - t0)
            dev_kernels.append(ctx.build_stats()["device_ms"])
        st = ctx.build_stats()
        row = {"triangles": n, "nodes": int(got_n.shape[0]), "levels": st["levels"], "launches": st["launches"],
               "device_kernels_ms": round(float(np.median(dev_kernels)), 2), "device_call_ms": round(1e3 * float(np.median(dev_wall)), 1)}
        # triangles in file order -> mesh ready to render (build + packing on the device, nodes back to the host)
        tex = np.zeros((2, 2, 4), np.uint8)
        ctx.set_option("device_build_min", 1)
        mid = ctx.upload_soup(soup, tex)
        t_re = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            ctx.rebuild_mesh(mid, soup)
            t_re.append(time.perf_counter() - t0)
        row["rt_mesh_rebuild_ms"] = round(1e3 * float(np.median(t_re)), 1)
        capi.lib().rt_scene_clear(ctx._h)
        if not a.no_host:
            t0 = time.perf_counter()
            want_t, want_n = capi.build_bvh(soup)
            row["host_call_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
            row["identical"] = bool(np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32)) and got_n.tobytes() == want_n.tobytes())
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps({"workload": "BVH build of random triangle soups, device vs host, both including the numpy copy of the soup",
                      "builds": rows}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
