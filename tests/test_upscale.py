"""The encoder's bicubic upscale (scene.c:188-202, utils/image.c:4-78): oracle and the product's strip
logic on CPU against golden vectors from the compiled reference; the HIP kernel through the C ABI on
the GPU.  Bar: every ARGB word identical."""
import hashlib
import os

import numpy as np
import pytest

import emul

NAMES = ("drone", "treasure", "ground")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden_up():
    from conftest import GOLDEN
    return dict(np.load(os.path.join(GOLDEN, "upscale.npz")))


@pytest.fixture(scope="module")
def golden_up_hashes():
    import json
    from conftest import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "upscale_hashes.json")))


def cases(golden_up, golden_frames):
    """(name, input rgb, sw, sh, expected) for every vector of upscale.npz."""
    for k in sorted(golden_up):
        if "_to_" not in k:
            continue
        src, dst = k.split("_to_")
        sw, sh = (int(v) for v in dst.split("x"))
        inp = golden_frames[src.split("_")[0] + "_rgb"] if src.startswith("f") else golden_up[src]
        yield k, inp, sw, sh, golden_up[k]


# ------------------------------------------------------------------------------------------ CPU

def test_scaled_size_rule(po):
    """scene.c:164-165: (int)(size / scale_factor + 0.5f)."""
    assert po.scaled_size(720, 540, 0.9) == (800, 600)          # the reference driver's configuration
    assert po.scaled_size(1920, 1080, 1.0) == (1920, 1080)
    assert po.scaled_size(1920, 1080, 0.9) == (2133, 1200)
    assert po.scaled_size(240, 135, 0.9) == (267, 150)
    from raytracer_amd import capi
    for w, h, s in ((720, 540, 0.9), (1920, 1080, 1.0), (1920, 1080, 0.9), (240, 135, 0.9), (7, 5, 0.3), (100, 100, 1.7)):
        assert capi.scaled_size(w, h, s) == po.scaled_size(w, h, s)
    with pytest.raises(capi.RtError):
        capi.scaled_size(10, 10, 0.0)


def test_oracle_upscale_equals_golden(po, golden_up, golden_frames):
    n = 0
    for k, inp, sw, sh, want in cases(golden_up, golden_frames):
        assert np.array_equal(po.upscale_frame(inp, sw, sh), want), k
        n += 1
    assert n >= 8


def test_oracle_upscale_full_size_hashes(po, meshes, golden_up_hashes):
    for key, want in golden_up_hashes.items():
        src, dst = key.split("_to_")
        size, f = src.split("_f")
        w, h = (int(v) for v in size.split("x"))
        sw, sh = (int(v) for v in dst.split("x"))
        fr = po.render_frame(meshes, po.animation_transforms(int(f)), po.default_camera(), po.default_light(), w, h, debug=False)
        assert sha(fr.rgb) == want["rgb_in"]
        assert sha(po.upscale_frame(fr.rgb, sw, sh)) == want["argb"], key


@pytest.mark.parametrize("rows", [1, 3, 8, 64])
def test_strip_logic_equals_golden(golden_up, golden_frames, rows):
    """The product's column strips (shared row results, window advance 0/1/2/3/many) for several strip heights."""
    for k, inp, sw, sh, want in cases(golden_up, golden_frames):
        assert np.array_equal(emul.upscale(inp, sw, sh, rows), want), (k, rows)


def test_strip_logic_equals_oracle_on_random_sizes(po):
    rng = np.random.default_rng(12)
    for _ in range(12):
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        sw, sh = int(rng.integers(2, 90)), int(rng.integers(2, 90))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        want = po.upscale_frame(img, sw, sh)
        assert (want >> 24 == 0xFF).all()
        assert np.array_equal(emul.upscale(img, sw, sh, int(rng.integers(1, 12))), want), (w, h, sw, sh)


def test_identity_scale_returns_the_frame(po, golden_frames):
    """scale_factor 1.0: source coordinates are whole numbers, the cubic returns the centre pixel."""
    img = golden_frames["f45_rgb"]
    out = po.upscale_frame(img, img.shape[1], img.shape[0])
    packed = (0xFF << 24) | (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
    assert np.array_equal(out, packed)


# ------------------------------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def gctx():
    from raytracer_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_upscale_equals_golden(gctx, golden_up, golden_frames):
    for k, inp, sw, sh, want in cases(golden_up, golden_frames):
        got = gctx.upscale_host(inp, sw, sh)
        assert np.array_equal(got, want), f"{k}: {(got != want).sum()} of {want.size} words differ"


@pytest.mark.gpu
def test_gpu_render_then_upscale_full_size(gctx, po, meshes, golden_up_hashes):
    """Frames rendered on the GPU, upscaled on the GPU without leaving HBM, hashes of the compiled reference."""
    import torch
    from raytracer_amd import capi
    for m in meshes:
        gctx.upload_mesh(m.tris, m.nodes, m.rgba)
    try:
        cam = capi.Camera(capi.Vec3(-3, 3, -3), capi.Vec3(0, 0, 0), capi.Vec3(0, 1, 0), 60.0)
        l = po.default_light()
        light = capi.Light(capi.Vec3(*l.direction.tup()), capi.Vec3(*l.color.tup()))
        dev = torch.device("cuda:0")
        for key, want in golden_up_hashes.items():
            src, dst = key.split("_to_")
            size, f = src.split("_f")
            w, h = (int(v) for v in size.split("x"))
            sw, sh = (int(v) for v in dst.split("x"))
            tfs = [capi.Transform(capi.Vec3(*t.position.tup()), capi.Vec3(*t.rotation.tup())) for t in po.animation_transforms(int(f))]
            # two copies of the frame in one batch, padded strides on both sides
            fs, os_ = w * h * 3 + 64, sw * sh + 32
            rgb = torch.zeros(2 * fs, dtype=torch.uint8, device=dev)
            argb = torch.full((2 * os_,), 0x01020304, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()          # the fills above ran on torch's stream, the launches below use another one
            stream = torch.cuda.Stream(device=dev)
            gctx.render_frames(cam, light, [tfs, tfs], w, h, rgb, frame_stride=fs, stream=stream.cuda_stream)
            gctx.upscale_frames(rgb, 2, w, h, sw, sh, argb, frame_stride=fs, out_stride=os_, stream=stream.cuda_stream)
            stream.synchronize()
            out = argb.cpu().numpy().view(np.uint32)
            assert sha(rgb[:w * h * 3].cpu().numpy()) == want["rgb_in"]
            for i in range(2):
                assert sha(out[i * os_:i * os_ + sw * sh]) == want["argb"], (key, i)
                assert (out[i * os_ + sw * sh:(i + 1) * os_] == 0x01020304).all()      # padding untouched
    finally:
        gctx.clear_scene()


@pytest.mark.gpu
def test_gpu_upscale_against_oracle_on_random_sizes(gctx, po):
    rng = np.random.default_rng(13)
    for _ in range(10):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        sw, sh = int(rng.integers(2, 400)), int(rng.integers(2, 300))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        assert np.array_equal(gctx.upscale_host(img, sw, sh), po.upscale_frame(img, sw, sh)), (w, h, sw, sh)


@pytest.mark.gpu
def test_gpu_upscale_argument_errors(gctx):
    from raytracer_amd import capi
    img = np.zeros((4, 4, 3), np.uint8)
    with pytest.raises(capi.RtError) as e:
        gctx.upscale_host(img, 1, 8)
    assert e.value.status == capi.RT_ERR_UNSUPPORTED
    import torch
    a = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    b = torch.zeros(64, dtype=torch.int32, device="cuda:0")
    with pytest.raises(capi.RtError):
        gctx.upscale_frames(a, 1, 4, 4, 8, 8, b, frame_stride=10)           # stride below one image
    with pytest.raises(capi.RtError):
        gctx.upscale_frames(a, 1, 4, 4, 8, 8, b, out_stride=10)
    with pytest.raises(capi.RtError):
        gctx.upscale_frames(a, 0, 4, 4, 8, 8, b)
