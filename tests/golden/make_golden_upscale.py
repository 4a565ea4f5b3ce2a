#!/usr/bin/env python3
"""Golden vectors of the encoder's bicubic upscale (scene.c:188-202 around image.c:25-78),
generated from the COMPILED REFERENCE (oracle/_ref/libref.so: the reference's utils/image.c as it is,
the two-line source-coordinate loop around it from oracle/ref_compose.c).

    make oracle && python tests/golden/make_golden_upscale.py

Writes upscale.npz: the ARGB result for two small frames of frames_240x135.npz (full arrays) and for
synthetic inputs at awkward sizes, and upscale_hashes.json: SHA-256 of the result at the reference
driver's sizes (720x540 -> 800x600, 1920x1080 -> 2133x1200 and -> 1920x1080).
"""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    po.ref()
    small = np.load(os.path.join(OUT, "frames_240x135.npz"))
    out = {}
    for f in (0, 95):
        sw, sh = po.scaled_size(240, 135, 0.9)
        out[f"f{f}_240x135_to_{sw}x{sh}"] = po.upscale_frame(small[f"f{f}_rgb"], sw, sh, use_ref=True)
    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)       # every pixel differs from its neighbours:
    out["noise_53x37"] = noise                                           # overshoot and clamping on both ends
    for sw, sh in ((53, 37), (64, 64), (211, 3), (2, 2), (40, 30)):
        out[f"noise_53x37_to_{sw}x{sh}"] = po.upscale_frame(noise, sw, sh, use_ref=True)
    tiny = rng.integers(0, 256, size=(1, 1, 3), dtype=np.uint8)
    out["tiny_1x1"] = tiny
    out["tiny_1x1_to_5x4"] = po.upscale_frame(tiny, 5, 4, use_ref=True)
    np.savez_compressed(os.path.join(OUT, "upscale.npz"), **out)

    # full sizes: frames rendered by the compiled reference
    soups = [po.load_obj(os.path.join(po.ASSETS, n + ".obj")) for n in ("drone", "treasure", "ground")]
    texs = [po.decode_webp(os.path.join(po.ASSETS, n + ".webp")) for n in ("drone", "treasure", "ground")]
    scene = po.RefScene(soups, texs)
    cam, light = po.default_camera(), po.default_light()
    hashes = {}
    for (w, h), scale, frame in (((720, 540), 0.9, 0), ((720, 540), 0.9, 45), ((1920, 1080), 0.9, 0), ((1920, 1080), 1.0, 0)):
        fr = scene.render(po.animation_transforms(frame), cam, light, w, h)
        sw, sh = po.scaled_size(w, h, scale)
        up = po.upscale_frame(fr.rgb, sw, sh, use_ref=True)
        hashes[f"{w}x{h}_f{frame}_to_{sw}x{sh}"] = {"argb": sha(up), "rgb_in": sha(fr.rgb)}
    scene.close()
    json.dump(hashes, open(os.path.join(OUT, "upscale_hashes.json"), "w"), indent=1, sort_keys=True)
    for fn in ("upscale.npz", "upscale_hashes.json"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


if __name__ == "__main__":
    main()
