#!/usr/bin/env python3
"""How long the leaf lists of a light-space grid would be, against what shadow rays cost today.

A directional light's shadow rays are parallel in a mesh's own space, so a 2-D grid across the light direction could name,
per cell, the reference leaves (accel/bvh.c:18-55: up to four triangles, split at the mean centroid along the longest axis) whose boxes
a ray through the cell can pass; the any-hit query would then test the listed leaf boxes instead of walking the tree.  This
script measures the lists such a grid would hold for the animation's meshes (raytracer.c:41-52) -- the footprint of a leaf
is the rectangle around its box's eight corners projected along the light -- for three grid sizes and three frames.

What to hold the numbers against: the box tests and triangle tests per shadow ray of the present walk (second tree,
nearer child first, stop at the first blocker), counted by the CPU build of the pixel function with the counters split
at the shadow loop: 5.7-6.3 box tests (3 of them the root tests of the three meshes, which the shadow rectangles remove in
the kernels) and 3.2-4.0 triangle tests per shadow ray at 1920x1080, frames 0 / 7 / 33.

    python scripts/shadow_grid_estimate.py          # CPU only; uses the restated BVH builder for the leaves
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pyoracle as po  # noqa: E402  (the leaves of the reference's tree: test infrastructure, nothing rendered here)


def leaf_lists(mesh, transform, light_dir, sizes=(32, 64, 128)):
    inv, _ = po.mesh_matrices(transform)
    d = inv[:3, :3].astype(np.float64) @ light_dir
    d /= np.linalg.norm(d)
    nodes = mesh.nodes
    leaves = nodes[(nodes["left"] < 0) & (nodes["right"] < 0)]
    bmin, bmax = leaves["bmin"].astype(np.float64), leaves["bmax"].astype(np.float64)
    k = int(np.argmax(np.abs(d)))
    a, b = (k + 1) % 3, (k + 2) % 3

    def across(p):   # the point where the line through p along d meets the plane x_k = 0
        return np.stack([p[:, a] - p[:, k] * d[a] / d[k], p[:, b] - p[:, k] * d[b] / d[k]], 1)

    corners = np.stack([across(np.where(np.array([cx, cy, cz]), bmax, bmin)) for cx in (0, 1) for cy in (0, 1) for cz in (0, 1)], 0)
    lo, hi = corners.min(0), corners.max(0)
    glo, ghi = lo.min(0), hi.max(0)
    rows = []
    for n in sizes:
        cell = (ghi - glo) / n
        i0 = np.floor((lo - glo) / cell).astype(int).clip(0, n - 1)
        i1 = np.floor((hi - glo) / cell).astype(int).clip(0, n - 1)
        count = np.zeros((n, n), int)
        for (x0, y0), (x1, y1) in zip(i0, i1):
            count[y0:y1 + 1, x0:x1 + 1] += 1
        used = count[count > 0]
        rows.append({"cells": n, "leaves": int(len(leaves)), "cells_used": round(len(used) / (n * n), 2), "mean_list": round(float(used.mean()), 1),
                     "p90_list": int(np.percentile(used, 90)), "longest_list": int(used.max()), "entries": int(count.sum())})
    return rows


def main():
    meshes = po.default_meshes()
    light = np.array(po.default_light().direction.tup(), np.float64)
    for frame in (0, 7, 33):
        transforms = po.animation_transforms(frame)
        for index in (0, 1):                                  # drone, treasure (the ground is flat and has its own grid)
            for row in leaf_lists(meshes[index], transforms[index], light):
                print({"frame": frame, "mesh": meshes[index].name, **row})


if __name__ == "__main__":
    main()
