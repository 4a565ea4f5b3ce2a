"""rt_render_frames_noalloc, the entry point that allocates nothing, against rt_render_frames through the C ABI:
same bytes on two calls in a row, the second one smaller (the static staging array keeps the first call's
contents), no device memory taken by the calls, and the refusals it documents."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_noalloc_frames_equal_rt_render_frames_on_two_calls_the_second_smaller():
    from raytracer_amd import capi, demo
    scene = demo.DemoScene(0)
    try:
        ctx = scene.ctx
        dev = torch.device("cuda", 0)
        cases = [([0, 45, 90, 17, 63], 320, 180), ([72, 5], 160, 90)]
        want, got = [], []
        for frames, w, h in cases:                       # outputs exist before the calls under test
            want.append(torch.zeros((len(frames), h, w, 3), dtype=torch.uint8, device=dev))
            got.append(torch.full((len(frames), h, w, 3), 7, dtype=torch.uint8, device=dev))
        # the allocating entry point first: it also puts the scene's node array on the device
        for (frames, w, h), out in zip(cases, want):
            ctx.render_frames(scene.camera, scene.light, [scene.transforms(f) for f in frames], w, h, out)
        ctx.sync()
        before = _free_bytes()
        for (frames, w, h), out in zip(cases, got):      # two calls in a row on the context's stream, the second smaller
            ctx.render_frames(scene.camera, scene.light, [scene.transforms(f) for f in frames], w, h, out, noalloc=True)
        ctx.sync()
        assert _free_bytes() == before, "device memory was taken by rt_render_frames_noalloc"
        for (frames, w, h), a, b in zip(cases, want, got):
            assert torch.equal(a, b), f"frames {frames} at {w}x{h} differ"
            assert int((a != 50).sum()) > 0              # not an all-background image
        # the largest batch the static array holds, on a caller's stream
        frames = list(range(0, 96, 6))
        assert len(frames) == capi.RT_MAX_BATCH
        a = torch.zeros((16, 72, 128, 3), dtype=torch.uint8, device=dev)
        b = torch.zeros_like(a)
        s = torch.cuda.Stream(device=dev)
        tf = [scene.transforms(f) for f in frames]
        ctx.render_frames(scene.camera, scene.light, tf, 128, 72, a, stream=s.cuda_stream)
        ctx.render_frames(scene.camera, scene.light, tf, 128, 72, b, stream=s.cuda_stream, noalloc=True)
        s.synchronize()
        assert torch.equal(a, b)
        # refusals: more frames than the array holds; a mesh list whose node array is not on the device yet
        with pytest.raises(capi.RtError) as e:
            ctx.render_frames(scene.camera, scene.light, tf + [tf[0]], 128, 72, torch.zeros((17, 72, 128, 3), dtype=torch.uint8, device=dev),
                              noalloc=True)
        assert e.value.status == capi.RT_ERR_CAPACITY and "RT_MAX_BATCH" in str(e.value)
        ctx.set_option("resident_budget", 1024)          # marks the scene as changed
        with pytest.raises(capi.RtError) as e:
            ctx.render_frames(scene.camera, scene.light, tf[:2], 128, 72, b[:2], noalloc=True)
        assert e.value.status == capi.RT_ERR_UNSUPPORTED
        ctx.render_frames(scene.camera, scene.light, tf[:2], 128, 72, b[:2])            # commits the scene again
        ctx.render_frames(scene.camera, scene.light, tf[:2], 128, 72, b[2:4], noalloc=True)
        ctx.sync()
        assert torch.equal(b[:2], b[2:4]) and torch.equal(b[:2], a[:2])
    finally:
        scene.close()
