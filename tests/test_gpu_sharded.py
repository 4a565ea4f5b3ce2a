"""The frame loop over several contexts of one process: rt_render_frames_sharded through the C ABI and
rt_host_set_gpus through the reference's host API.  The GPU box has one GPU, so "several GPUs" is rehearsed with
several contexts on device 0 -- the dealing of frames, the staging slots and the copies to the gathering device
are the code an 8-GPU node runs; what is not covered here is the xGMI transfer itself (hardware-unmeasured)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_sharded_frame_loop_equals_the_batched_one():
    import torch
    from raytracer_amd import capi, demo
    w, h = 160, 90
    scenes = [demo.DemoScene(0) for _ in range(3)]
    try:
        cam, light = scenes[0].camera, scenes[0].light
        assert capi.device_count() >= 1
        for n_frames in (1, 16, 40, 53):
            tfs = [scenes[0].transforms(f) for f in range(n_frames)]
            want = torch.zeros((n_frames, h, w, 3), dtype=torch.uint8, device="cuda:0")
            for b in range(0, n_frames, 16):
                scenes[0].ctx.render_frames(cam, light, tfs[b:b + 16], w, h, want[b:b + 16])
            scenes[0].ctx.sync()
            for n_ctx in (1, 2, 3):
                got = torch.full((n_frames, h, w, 3), 9, dtype=torch.uint8, device="cuda:0")
                capi.render_frames_sharded([s.ctx for s in scenes[:n_ctx]], cam, light, tfs, w, h, got)
                assert torch.equal(got, want), f"{n_frames} frames over {n_ctx} contexts"
        # a padded frame stride, and nothing to do
        tfs = [scenes[0].transforms(f) for f in range(7)]
        stride = w * h * 3 + 256
        flat = torch.zeros(7 * stride, dtype=torch.uint8, device="cuda:0")
        capi.render_frames_sharded([s.ctx for s in scenes[:2]], cam, light, tfs, w, h, flat, frame_stride=stride)
        for f in range(7):
            assert torch.equal(flat[f * stride:f * stride + w * h * 3].reshape(h, w, 3), want[f])
            assert int(flat[f * stride + w * h * 3:(f + 1) * stride].sum()) == 0
        capi.render_frames_sharded([scenes[0].ctx], cam, light, [], w, h, flat)
        # refusals: the same context twice, a context without the scene
        with pytest.raises(capi.RtError):
            capi.render_frames_sharded([scenes[0].ctx, scenes[0].ctx], cam, light, tfs, w, h, flat, frame_stride=stride)
        with capi.Context(0) as empty:
            with pytest.raises(capi.RtError):
                capi.render_frames_sharded([scenes[0].ctx, empty], cam, light, tfs, w, h, flat, frame_stride=stride)
    finally:
        for s in scenes:
            s.close()


def test_host_api_deals_frames_over_a_gpu_list():
    """rt_host_set_gpus: the reference driver's loop with frame f on list entry f mod n (two entries, both device 0, four
    frames in flight); the first 12 frames of the animation equal the compiled reference's, hash for hash."""
    from raytracer_amd import host_api as H
    hashes = json.load(open(os.path.join(GOLDEN, "animation_hashes.json")))
    L = H.lib()
    assert L.rt_host_set_gpus((C.c_int * 2)(0, 99), 2) != 0            # no such device
    assert L.rt_host_set_gpus((C.c_int * 2)(0, 0), 2) == 0
    L.rt_host_set_frames_in_flight(4)
    try:
        scene = L.create_scene(800, 600, 4000, 24, 0.9)
        L.set_scene_camera(C.byref(scene), H.Vec3(-3, 3, -3), H.Vec3(0, 0, 0), H.Vec3(0, 1, 0), 60.0)
        L.set_scene_light(C.byref(scene), H.Vec3(1, 1, -1), H.Vec3(1.4, 1.4, 1.4))
        loaded = []
        for n in ("drone", "treasure", "ground"):
            m = L.create_mesh(os.path.join(H.ASSETS, n + ".obj").encode(), os.path.join(H.ASSETS, n + ".webp").encode())
            loaded.append(m)
            L.add_mesh_to_scene(C.byref(scene), m)
        for frame in range(12):
            drone, treasure = H.Transform(), H.Transform()
            L.rt_host_demo_animation(frame, C.byref(drone), C.byref(treasure))
            L.set_mesh_position(C.byref(scene.meshes[0]), drone.position)
            L.set_mesh_rotation(C.byref(scene.meshes[0]), drone.rotation)
            L.set_mesh_position(C.byref(scene.meshes[1]), treasure.position)
            L.set_mesh_rotation(C.byref(scene.meshes[1]), treasure.rotation)
            L.render_scene(C.byref(scene))
            assert L.rt_host_last_status(C.byref(scene)) == 0
            L.next_frame(C.byref(scene))
        assert L.rt_host_wait(C.byref(scene)) == 0
        assert (scene.width, scene.height) == (hashes["width"], hashes["height"])
        for frame in range(12):
            assert hashlib.sha256(H.frame_array(scene, frame).tobytes()).hexdigest() == hashes["frames"][frame], f"frame {frame}"
        for m in loaded:
            L.destroy_mesh(C.byref(m))
        L.destroy_scene(C.byref(scene))
    finally:
        L.rt_host_set_frames_in_flight(1)
        L.rt_host_set_gpus(None, 0)


def _animation_job(*args):
    import subprocess
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "animation_job.py"), *args], capture_output=True, text=True,
                          timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"job"' in ln]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_the_whole_animation_as_one_job_equals_the_compiled_reference(tmp_path):
    """The reference driver's 96 frames at its own size, rendered in batches and handed to the animation writer: the hash
    over all frames is the compiled reference's, and the file holds 96 frames."""
    out = str(tmp_path / "job.webp")
    got = _animation_job("--out", out)
    assert got["frames"] == 96 and got["equals_compiled_reference"] is True
    assert os.path.getsize(out) > 100_000


@pytest.mark.gpu
def test_the_animation_job_over_two_ranks_gathers_the_same_frames():
    """Two ranks on the one GPU, frames gathered through gloo: rank 0 ends up with the same 96 frames in order."""
    got = _animation_job("--gpus", "2", "--backend", "gloo", "--one-device")
    assert got["n_gpus"] == 2 and got["frames"] == 96 and got["equals_compiled_reference"] is True


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


@pytest.mark.gpu
def test_the_animation_job_through_rccl_with_a_single_rank():
    """One rank under torch.distributed.run with the nccl backend: the process group is RCCL's and every batch goes through its
    gather (to itself).  What a one-GPU machine can show of the path N GPUs take."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                           "--master-port", _free_port(), os.path.join(repo, "scripts", "animation_job.py"), "--backend", "nccl",
                           "--collective-at-one-rank"], capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    got = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{") and '"job"' in ln]
    assert len(got) == 1 and got[0]["backend"] == "nccl" and got[0]["frames"] == 96 and got[0]["equals_compiled_reference"] is True


@pytest.mark.gpu
def test_bench_through_rccl_with_a_single_rank():
    """bench.py as one rank of an RCCL process group: the gathers of the timed region and the gathered-bytes check run through
    RCCL's gather."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                           "--master-port", _free_port(), os.path.join(repo, "bench.py"), "--collective-at-one-rank", "--steps", "16", "--warmup", "8",
                           "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["backend"] == "nccl" and out["config"]["world_size"] == 1
    assert out["config"]["gathered_frames_equal_rank0_render"] is True and out["value"] > 0
