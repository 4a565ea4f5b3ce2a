"""Frame sharding and the framebuffer gather, on CPU tensors with the gloo backend (two and three
processes).  The frames are produced by the CPU oracle here -- the subject of the test is the
distribution logic, which is the same code that runs over RCCL on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import REPO


def test_frames_for_rank_partitions_the_animation():
    from raytracer_amd.sharding import frames_for_rank, owner_of
    for n in (0, 1, 7, 96):
        for world in (1, 2, 3, 8):
            shares = [frames_for_rank(n, r, world) for r in range(world)]
            assert sorted(f for s in shares for f in s) == list(range(n))
            assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
            for r, s in enumerate(shares):
                assert all(owner_of(f, world) == r for f in s)
    with pytest.raises(ValueError):
        frames_for_rank(10, 2, 2)
    with pytest.raises(ValueError):
        frames_for_rank(10, 0, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, batch, w, h, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as po
    from raytracer_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    meshes = po.default_meshes()
    cam, light = po.default_camera(), po.default_light()
    rendered = []

    def render_one(frame, out):
        fr = po.render_frame(meshes, po.animation_transforms(frame), cam, light, w, h, threads=2, debug=False)
        out.copy_(torch.from_numpy(fr.rgb))
        rendered.append(frame)

    frames = sharding.render_sharded(n_frames, rank, world, h, w, render_one, torch.device("cpu"), batch=batch)
    assert rendered == sharding.frames_for_rank(n_frames, rank, world)
    if rank == 0:
        assert sorted(frames) == list(range(n_frames))
        np.save(os.path.join(out_dir, "gathered.npy"), np.stack([frames[f].numpy() for f in range(n_frames)]))
    else:
        assert frames is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,batch", [(2, 7, 2), (2, 4, 4), (3, 5, 1)])
def test_sharded_render_gathers_every_frame_on_rank0(tmp_path, po, meshes, world, n_frames, batch):
    import torch.multiprocessing as mp
    w, h = 48, 27
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, batch, w, h, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish")
        assert p.exitcode == 0
    got = np.load(os.path.join(tmp_path, "gathered.npy"))
    assert got.shape == (n_frames, h, w, 3)
    for f in range(n_frames):
        want = po.render_frame(meshes, po.animation_transforms(f), po.default_camera(), po.default_light(), w, h, debug=False)
        assert np.array_equal(got[f], want.rgb), f"frame {f}"


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from raytracer_amd.sharding import FrameGather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FrameGather(rank, world)
    like = torch.zeros((2, 4, 5, 3), dtype=torch.uint8)
    g.prealloc(3, like)
    sent = []
    for i in range(3):                                  # several submits in flight before finish()
        t = torch.full((2, 4, 5, 3), 10 * rank + i, dtype=torch.uint8)
        sent.append(t)
        g.submit(t)
    got = g.finish()
    if rank == 0:
        assert len(got) == 3
        for i, per in enumerate(got):
            assert len(per) == world
            for r, t in enumerate(per):
                assert (t == 10 * r + i).all()
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_frame_gather_keeps_submit_order_with_several_in_flight(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert os.path.exists(os.path.join(tmp_path, "ok"))


def _ring_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from raytracer_amd.sharding import FrameGather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def consume(seq, per_rank):
        seen.append((seq, [int(t.flatten()[0]) for t in per_rank]))

    g = FrameGather(rank, world, ring=2, consume=consume)
    like = torch.zeros((2, 3, 4, 3), dtype=torch.uint8)
    g.prealloc(7, like)
    send = [torch.zeros_like(like) for _ in range(3)]            # ring of send buffers, one more than gathers in flight
    for i in range(7):
        send[i % 3].fill_(10 * rank + i)
        g.submit(send[i % 3])
        assert len(g._pending) <= 2
    assert g.finish() is None
    if rank == 0:
        assert seen == [(i, [10 * r + i for r in range(world)]) for i in range(7)], seen
        assert len(g._pool) == 2                                 # receive buffers were recycled, not re-allocated
        open(os.path.join(out_dir, "ring_ok"), "w").write("ok")
    else:
        assert seen == []
    dist.barrier()
    dist.destroy_process_group()


def test_frame_gather_ring_bounds_what_is_in_flight(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert os.path.exists(os.path.join(tmp_path, "ring_ok"))


def test_frame_gather_single_rank_is_a_passthrough():
    import torch
    from raytracer_amd.sharding import FrameGather
    g = FrameGather(0, 1)
    t = torch.arange(24, dtype=torch.uint8).reshape(1, 2, 4, 3)
    g.submit(t)
    out = g.finish()
    assert len(out) == 1 and out[0][0] is t


# ------------------------------------------------------------------ one frame over several ranks
@pytest.mark.parametrize("height", [1, 7, 8, 9, 64, 135, 540, 1080])
@pytest.mark.parametrize("world", [1, 2, 3, 8, 200])
def test_strip_shares_tile_the_frame(height, world):
    """Every row of the frame belongs to exactly one rank's share, and the C library counts the same rows."""
    from raytracer_amd import capi, sharding
    seen = torch.zeros(height, dtype=torch.int64)
    for r in range(world):
        rows = sharding.frame_rows_of_share(height, world, r)
        assert rows.numel() == sharding.strip_rows(height, world, r) == capi.strip_rows(height, world, r)
        assert rows.numel() == 0 or (rows.max() < height and (rows[1:] > rows[:-1]).all())
        seen[rows] += 1
    assert (seen == 1).all()
    frame = torch.arange(height * 5, dtype=torch.int32).reshape(height, 5)
    shares = [frame[sharding.frame_rows_of_share(height, world, r)] for r in range(world)]
    padded = [torch.cat([s, torch.full((3, 5), -1, dtype=torch.int32)]) for s in shares]          # padding rows are ignored
    assert torch.equal(sharding.assemble_strips(padded, height), frame)


def _gather_frame_worker(rank, world, port, height, q):
    import torch.distributed as dist
    from raytracer_amd import sharding
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    frame = (torch.arange(height * 4 * 3, dtype=torch.int64) % 251).to(torch.uint8).reshape(height, 4, 3)
    share = frame[sharding.frame_rows_of_share(height, world, rank)].contiguous()        # what this rank would have rendered
    got = sharding.gather_frame(share, height, rank, world)
    if rank == 0:
        q.put(bool(torch.equal(got, frame)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height", [135, 13])
def test_gather_frame_two_ranks_gloo(height):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_frame_worker, args=(r, 2, port, height, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


# ------------------------------------------------------------------ bench.py as a plain command
def _run_bench(*args, timeout=240):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(args), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)


def test_bench_starts_its_own_ranks_when_given_a_gpu_count():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the program starts two ranks itself
    (here on byte patterns over gloo, no GPU), rank 0's JSON line comes back through the parent."""
    import json
    r = _run_bench("--gpus", "2", "--rehearse-cpu", "--steps", "8", "--warmup", "2", "--width", "64", "--height", "36", "--batch", "4")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2 and out["config"]["backend"] == "gloo"
    assert out["config"]["rank_devices"] == ["cpu", "cpu"]
    assert out["config"]["gathered_frames_equal_rank0_render"] is True
    assert "rehearsal" in out and out["steps"] == 8 and out["scaling"] == "weak"


def test_bench_eight_ranks_as_the_first_multi_gpu_command_will_run():
    """`python bench.py --gpus 8 --steps 20 --warmup 5` at the bench's own frame size, rehearsed on byte patterns over gloo: eight
    ranks report, every rank deals itself 12 frames (batches of 7 + 7 + 6 per repetition, a ring of three gathers in flight), and
    rank 0 checks every pixel of every frame of every gathered batch against the byte its frame number gives it."""
    import json
    r = _run_bench("--gpus", "8", "--rehearse-cpu", "--steps", "20", "--warmup", "5", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["config"]["world_size"] == 8 and out["config"]["rank_devices"] == ["cpu"] * 8
    assert out["config"]["frames_per_launch"] == 7 and out["config"]["launch_sizes"] == [7, 7, 6]
    assert out["config"]["gathered_frames_equal_rank0_render"] is True
    reps = out["timing"]["repeats"]
    assert reps >= 3 and out["config"]["rehearsal_frames_checked_on_rank0"] == 8 * 20 * reps
    assert out["config"]["rehearsal_frames_right"] == out["config"]["rehearsal_frames_checked_on_rank0"]
    assert out["gather_budget"]["gather_bytes_per_step"] == 1920 * 1080 * 3
    lo, med, hi = out["timing"]["ms_per_step_min_median_max"]
    assert lo <= med <= hi and abs(out["ms_per_step"] - med) < 1e-3


def test_animation_job_over_eight_ranks_delivers_every_frame_in_order():
    """scripts/animation_job.py --gpus 8 (the reference driver's 96 frames, frame f on rank f mod 8, two gather messages of six
    frames per rank, the last one padded), rehearsed on byte patterns over gloo: rank 0 ends up with frames 0..95 in order."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "animation_job.py"), "--gpus", "8", "--rehearse-cpu", "--width", "720", "--height", "540"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["backend"] == "gloo" and out["frames"] == 96 and out["frames_verified_in_order"] == 96
    assert out["frames_per_launch"] == 6


def test_bench_single_rank_can_go_through_the_process_group():
    """One rank under torch.distributed.run with --collective-at-one-rank: the process group exists and every batch is gathered
    through it (here gloo on byte patterns; on a GPU machine the same switch runs RCCL's gather with one GPU)."""
    import json
    import subprocess
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--rehearse-cpu", "--collective-at-one-rank", "--steps", "8",
                        "--warmup", "2", "--width", "64", "--height", "36", "--batch", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["world_size"] == 1 and out["config"]["backend"] == "gloo"
    assert out["config"]["gathered_frames_equal_rank0_render"] is True


def test_bench_parent_fails_when_a_rank_fails():
    """Without a GPU the ranks refuse to run (there is no CPU rendering path); the parent must not exit 0."""
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--steps", "8", "--warmup", "0")
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_bench_frame_sequence_does_not_depend_on_the_step_count():
    sys.path.insert(0, REPO)
    import bench
    from raytracer_amd.sharding import frames_for_rank
    for world in (1, 2, 4, 8):
        for rank in range(world):
            share = frames_for_rank(96, rank, world)
            full = bench.frame_sequence(share, len(share))
            assert sorted(full) == share                                  # one pass visits every frame of the share once
            for n in (5, 20, 96):
                assert bench.frame_sequence(share, n) == bench.frame_sequence(share, 200)[:n]
    # twenty steps of one GPU spread over the whole animation, not its first quarter
    twenty = bench.frame_sequence(list(range(96)), 20)
    assert min(twenty) < 10 and max(twenty) > 85 and len(set(f // 24 for f in twenty)) == 4
