"""The animation writer behind save_scene (reference scene.c:164-224).

The reference's encoder output is lossy and comes from libwebpmux, which is not available, so there
is no byte-level reference output to pin (parity unpinned for the file's bytes).  What is checked:
the container is read back by an independent decoder (Pillow's own libwebp with its demuxer), frame
count, canvas, loop count, background and frame times are the ones the reference asks its encoder
for, and -- in lossless mode -- every decoded frame equals the frame that went in, bit for bit.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def L():
    from raytracer_amd import host_api as H
    return H.lib()


def frames_of(n, w, h, seed=1):
    rng = np.random.default_rng(seed)
    base = (rng.random((h, w, 3)) * 255).astype(np.uint8)
    base[: h // 2] = np.linspace(0, 255, w, dtype=np.uint8)[None, :, None]        # something an encoder can compress
    return [np.ascontiguousarray(np.roll(base, 5 * k, axis=1)) for k in range(n)]


def write(L, path, frames, durations, quality, threads, argb=False):
    h, w = frames[0].shape[:2]
    a = C.c_void_p()
    assert L.rt_host_animation_open(path.encode(), w, h, quality, threads, C.byref(a)) == 0
    for f, d in zip(frames, durations):
        if argb:
            words = (0xFF000000 | (f[..., 0].astype(np.uint32) << 16) | (f[..., 1].astype(np.uint32) << 8) | f[..., 2]).astype(np.uint32)
            assert L.rt_host_animation_add(a, words.ctypes.data, 1, d) == 0
        else:
            assert L.rt_host_animation_add(a, f.ctypes.data, 0, d) == 0
    assert L.rt_host_animation_close(a) == 0


def read(path):
    im = Image.open(path)
    out, durations = [], []
    for k in range(getattr(im, "n_frames", 1)):
        im.seek(k)
        out.append(np.asarray(im.convert("RGB")).copy())
        durations.append(im.info.get("duration"))
    return im, out, durations


@pytest.mark.parametrize("threads,argb", [(1, False), (4, True), (0, False)])
def test_lossless_animation_round_trip(L, tmp_path, threads, argb):
    assert L.rt_host_animation_has_webp() == 1
    n, w, h = 23, 161, 95                        # odd sizes: chunk padding; more frames than queue slots with one thread
    frames = frames_of(n, w, h)
    # scene.c:204,209: timestamps frame*(duration/frame_count), the end at duration_ms
    duration_ms, step = 1000, 1000 // n
    durations = [step] * (n - 1) + [duration_ms - (n - 1) * step]
    path = str(tmp_path / "a.webp")
    write(L, path, frames, durations, -1.0, threads, argb)
    im, got, got_d = read(path)
    assert im.size == (w, h) and im.n_frames == n
    assert im.info["loop"] == 0 and tuple(im.info["background"]) == (255, 255, 255, 255)
    assert got_d == durations
    for k in range(n):
        assert np.array_equal(got[k], frames[k]), k
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and int.from_bytes(raw[4:8], "little") == len(raw) - 8 and raw[8:16] == b"WEBPVP8X"


def test_lossy_animation_is_close(L, tmp_path):
    frames = frames_of(6, 320, 200)
    for f in frames:
        f[100:] = 90
    a, b = str(tmp_path / "lossy.webp"), str(tmp_path / "lossless.webp")
    write(L, a, frames, [41] * 6, 75.0, 2)
    write(L, b, frames, [41] * 6, -1.0, 2)
    _, got, _ = read(a)
    assert len(got) == 6
    for k in range(6):
        err = np.abs(got[k][100:].astype(int) - frames[k][100:].astype(int))
        assert err.mean() < 3.0
    assert b"VP8 " in open(a, "rb").read()[:200] and b"VP8L" in open(b, "rb").read()[:200]


def test_single_frame_and_bad_arguments(L, tmp_path):
    f = frames_of(1, 2, 2)
    path = str(tmp_path / "one.webp")
    write(L, path, f, [4000], -1.0, 1)
    im, got, d = read(path)
    assert np.array_equal(got[0], f[0]) and d == [4000]
    a = C.c_void_p()
    assert L.rt_host_animation_open(path.encode(), 0, 10, 75.0, 1, C.byref(a)) < 0
    assert L.rt_host_animation_open(path.encode(), 10, 20000, 75.0, 1, C.byref(a)) < 0
    assert L.rt_host_animation_open(str(tmp_path / "no" / "dir.webp").encode(), 4, 4, 75.0, 1, C.byref(a)) < 0 and not a.value
    assert L.rt_host_animation_add(None, f[0].ctypes.data, 0, 1) < 0
    assert L.rt_host_animation_close(None) < 0


def test_ppm_stream_without_an_encoder(tmp_path):
    """With no WebP encoder in the process the same calls write concatenated P6 images."""
    code = f"""
import sys; sys.path.insert(0, {REPO!r}); sys.path.insert(0, {os.path.join(REPO, 'tests')!r})
import ctypes as C, numpy as np
from raytracer_amd import host_api as H
L = H.lib()
assert L.rt_host_animation_has_webp() == 0
a = C.c_void_p()
assert L.rt_host_animation_open({str(tmp_path / 'x.webp')!r}.encode(), 5, 3, 75.0, 2, C.byref(a)) == 0
for k in range(4):
    f = np.full((3, 5, 3), k * 10, np.uint8)
    assert L.rt_host_animation_add(a, f.ctypes.data, 0, 41) == 0
assert L.rt_host_animation_close(a) == 0
"""
    env = dict(os.environ, RT_HOST_NO_WEBP="1")
    subprocess.run([sys.executable, "-c", code], check=True, env=env)
    raw = open(str(tmp_path / "x.webp.ppm"), "rb").read()
    head = b"P6\n5 3\n255\n"
    assert len(raw) == 4 * (len(head) + 45)
    for k in range(4):
        part = raw[k * (len(head) + 45):(k + 1) * (len(head) + 45)]
        assert part[:len(head)] == head and part[len(head):] == bytes([k * 10]) * 45


def test_frame_times_follow_the_reference():
    """scene.c:204,209 with the driver's 4000 ms / 96 frames: 41 ms steps, the last frame lasts to 4000."""
    from raytracer_amd import animation
    t = animation.frame_times(96, 4000)
    assert t[:95] == [41] * 95 and t[95] == 4000 - 95 * 41 and sum(t) == 4000
    assert animation.frame_times(0, 100) == [] and animation.frame_times(1, 100) == [100]
