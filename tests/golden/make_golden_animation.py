#!/usr/bin/env python3
"""SHA-256 of every frame of the reference driver's animation (raytracer.c:10-57: 96 frames, 720x540)
as the COMPILED REFERENCE renders them (oracle/_ref/libref.so, see make_golden.py for what that is),
plus one hash over all of them.  Data only.

    make oracle && python tests/golden/make_golden_animation.py
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NAMES = ("drone", "treasure", "ground")


def main():
    R = po.ref()
    soups = [po.load_obj(os.path.join(po.ASSETS, n + ".obj")) for n in NAMES]
    texs = [po.decode_webp(os.path.join(po.ASSETS, n + ".webp")) for n in NAMES]
    scene = po.RefScene(soups, texs)
    cam = po.default_camera()
    light = po.Light()
    d, c = po.vec3(1, 1, -1), po.vec3(1.4, 1.4, 1.4)
    R.refc_make_light(C.byref(d), C.byref(c), C.byref(light))
    w, h, n = 720, 540, 96
    frames, total = [], hashlib.sha256()
    for f in range(n):
        fr = scene.render(po.animation_transforms(f), cam, light, w, h)
        frames.append(hashlib.sha256(fr.rgb.tobytes()).hexdigest())
        total.update(fr.rgb.tobytes())
        print(f, frames[-1][:16], flush=True)
    json.dump({"width": w, "height": h, "frames": frames, "all_frames_sha256": total.hexdigest()},
              open(os.path.join(OUT, "animation_hashes.json"), "w"), indent=1)
    scene.close()


if __name__ == "__main__":
    main()
