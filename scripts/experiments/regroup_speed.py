#!/usr/bin/env python3
"""Experiment: the reference's leaves kept as they are, the tree above them rebuilt with a binned SAH.
Times rt_render_frames on both forms of the scene and compares the frames.

    python scripts/experiments/regroup_speed.py
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.setrecursionlimit(100000)


def regroup(nodes, dtype, bins=16):
    leaves = [i for i in range(len(nodes)) if nodes[i]["left"] < 0]
    lo = np.array([nodes[i]["bmin"] for i in leaves]); hi = np.array([nodes[i]["bmax"] for i in leaves])
    cen = (lo + hi) * 0.5
    out = []

    def area(l, h):
        d = np.maximum(h - l, 0)
        return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])

    def rec(idx):
        me = len(out); out.append(None)
        if len(idx) == 1:
            n = nodes[leaves[idx[0]]]
            out[me] = (n["bmin"].copy(), n["bmax"].copy(), int(n["start"]), int(n["count"]), -1, -1)
            return me
        bl = lo[idx].min(0); bh = hi[idx].max(0)
        best = (np.inf, None, None)
        cl = cen[idx].min(0); ch = cen[idx].max(0)
        for ax in range(3):
            if ch[ax] - cl[ax] <= 0:
                continue
            b = np.minimum(((cen[idx, ax] - cl[ax]) / (ch[ax] - cl[ax]) * bins).astype(int), bins - 1)
            for s in range(1, bins):
                L = idx[b < s]; R = idx[b >= s]
                if len(L) == 0 or len(R) == 0:
                    continue
                c = area(lo[L].min(0), hi[L].max(0)) * len(L) + area(lo[R].min(0), hi[R].max(0)) * len(R)
                if c < best[0]:
                    best = (c, ax, s)
        if best[1] is None:
            L = idx[:len(idx) // 2]; R = idx[len(idx) // 2:]
        else:
            ax, s = best[1], best[2]
            b = np.minimum(((cen[idx, ax] - cl[ax]) / (ch[ax] - cl[ax]) * bins).astype(int), bins - 1)
            L = idx[b < s]; R = idx[b >= s]
        l = rec(L); r = rec(R)
        out[me] = (bl, bh, 0, 0, l, r)
        return me

    rec(np.arange(len(leaves)))
    arr = np.zeros(len(out), dtype)
    for i, (bl, bh, s, c, l, r) in enumerate(out):
        arr[i]["bmin"] = bl; arr[i]["bmax"] = bh; arr[i]["start"] = s; arr[i]["count"] = c; arr[i]["left"] = l; arr[i]["right"] = r
    return arr


def main():
    from raytracer_amd import capi, demo, host_api
    from oracle import pyoracle as po
    dev = torch.device("cuda", 0)
    scene = demo.DemoScene(0)
    meshes = po.default_meshes()
    ctxs = {"reference tree": capi.Context(0), "regrouped": capi.Context(0)}
    for m in meshes:
        ctxs["reference tree"].upload_mesh(m.tris, m.nodes, m.rgba)
        ctxs["regrouped"].upload_mesh(m.tris, regroup(m.nodes, capi.NODE_DTYPE), m.rgba)
    W, H = 1920, 1080
    out = {}
    bufs = {}
    for name, c in ctxs.items():
        buf = torch.empty((8, H, W, 3), dtype=torch.uint8, device=dev)
        ms = []
        for rep in range(3):
            for b in range(12):
                fr = list(range(b * 8, b * 8 + 8))
                c.render_frames(scene.camera, scene.light, [scene.transforms(f) for f in fr], W, H, buf)
                c.sync()
                if rep:
                    ms.append(c.last_kernel_ms())
        out[name] = {"ms_per_frame": round(float(np.mean(ms)) / 8, 4)}
        bufs[name] = buf
    out["last_batch_identical"] = bool(torch.equal(bufs["reference tree"], bufs["regrouped"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
