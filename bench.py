#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s (primary + shadow) on the reference's drone / treasure /
ground scene at 1920x1080, one ray per pixel plus one shadow ray per hit pixel.

    python bench.py --gpus N --steps K --warmup W

A step is one frame.  Rank r of N owns frames r, r+N, r+2N, ... of the 96-frame animation (per-GPU
work is fixed as N grows: weak scaling).  Step k of a rank renders entry (7 k) mod L of its L own
frames: 7 is coprime to every share length, so ANY number of steps is an even sample of the whole
animation and the value does not depend on --steps or --warmup (frames differ by 2x in cost).
Frames are submitted `--batch` at a time through rt_render_frames (one launch per batch: the
long-running tiles of every frame in the batch overlap with the bulk of the others); for N > 1 each
finished batch of RGB8 framebuffers is gathered to rank 0 (torch.distributed: RCCL with the default
--backend nccl) inside the timed region while the next batch renders.  Framebuffers stay in HBM;
nothing is copied to the host inside the timed region.  Before the W warm-up steps every rank keeps
its GPU busy with 24 untimed launches, so that the short timed region runs at working clocks.
The timed region -- exactly K steps between barrier + synchronize on both sides -- lasts a few
milliseconds, so it is repeated (the same K steps each time, at least five times and until half a
second has gone by); `value` comes from the median repetition (maximum over ranks per repetition),
and the spread is reported beside it (`timing`).

With --gpus N > 1 and no WORLD_SIZE in the environment this program starts its own N ranks (a
child `python -m torch.distributed.run`, before anything in this process has touched a GPU) and
relays rank 0's JSON line; under torch.distributed.run it is one of the ranks.

A ray is one camera ray or one shadow ray tested against the whole scene (all meshes):
rays per frame = pixels + pixels that hit something.

Rank 0 prints one JSON line.  Besides the throughput it carries
  roofline      the render kernel against the roof that binds it, FP32 vector issue: algorithmic
                FLOPs of the reference's traversal (22 per box test, 50 per triangle test, 45 per
                ray and mesh for the transform) over the kernel's mean duration measured with HIP
                events on its stream, against the 157.3 TFLOP/s vector peak.  `traffic` is HBM bytes
                per launch from a separate rocprofv3 --pmc pass (profiles/), `hbm_frac` the
                resulting share of the 8 TB/s HBM peak; the algorithmic bytes are served by L1/L2.
  cpu_baseline  the compiled reference functions (or, without them, the CPU restatement) timed on
                the host cores on a bounded sample of the same workload (N = 1, rank 0 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3       # MI355X FP32 vector peak (packed FMA on every lane)
# Algorithmic bytes / flops of the reference's traversal.
B_SLAB, B_TRI, B_HIT, B_PIXEL = 32, 36, 64, 3
F_SLAB, F_TRI, F_XFORM = 22, 50, 45
FRAME_STRIDE = 7               # step k renders entry (7 k) mod L of the rank's L frames
SETTLE = 24                    # untimed launches before the warm-up
PMC_PROFILE = os.path.join(REPO, "profiles", "r3_traffic.json")
ISSUE_PROFILE = os.path.join(REPO, "profiles", "r3_valu_issue_summary.json")
XGMI_LINK_GBS_ONE_WAY = 76.8   # MI355X: seven links of 153.6 GB/s (both directions together) per GPU, one per peer
MIN_TIMED_SECONDS, MIN_REPEATS, MAX_REPEATS = 0.5, 5, 2000


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=96)
    p.add_argument("--warmup", type=int, default=16)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--batch", type=int, default=0,
                   help="frames per launch (and per gather message for N > 1); 0: FRAMES_PER_LAUNCH on one GPU, half of it on several "
                        "(shorter messages: more of the gather travels behind the next launch)")
    p.add_argument("--block", type=int, default=0)
    p.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                   help="library option for the rendering context (rt_set_option), e.g. batch_chunk=480; may be given several times")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--verbose", action="store_true", help="phase markers on stderr")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    p.add_argument("--collective-at-one-rank", action="store_true",
                   help="one rank started under torch.distributed.run: create the process group and gather through it all the same "
                        "(the backend's calls on a one-GPU machine)")
    p.add_argument("--one-device", action="store_true",
                   help="rehearsal on a one-GPU machine: every rank uses GPU 0 (use with --backend gloo)")
    p.add_argument("--rehearse-cpu", action="store_true",
                   help="no GPU and no rendering: frames are byte patterns in host memory (forces --backend gloo).  "
                        "Exercises rank start-up, frame dealing, the gathers and their check; its value is not a measurement")
    p.add_argument("--cpu-frames", type=int, default=48, help="animation frames the CPU baseline renders (an even sample; 48 take 10-12 s on 16 threads)")
    p.add_argument("--master-port", type=int, default=0, help="rendezvous port when this program starts its own ranks (0 = a free one)")
    p.add_argument("--repeats", type=int, default=0, help="repetitions of the timed region (0: at least five, and until half a second has gone by)")
    return p.parse_args(argv)


FRAMES_PER_LAUNCH = 16        # RT_MAX_BATCH: the tail of a launch is shared by more frames (8 per launch: 5 % slower)


def frame_sequence(share, n):
    """Frames of steps 0..n-1 of a rank that owns `share`."""
    return [share[(k * FRAME_STRIDE) % len(share)] for k in range(n)]


def spawn_ranks(a, argv, script=None, marker='"metric"'):
    """Starts N ranks of this program (or of `script`) under torch.distributed.run as a child process and relays
    rank 0's JSON line (the stdout line that holds `marker`).  Nothing in this process has initialised a GPU
    (torch is not even imported here), and the children are fresh interpreters."""
    port = a.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script or __file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and marker in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        sys.exit(f"{os.path.basename(script or __file__)}: the {a.gpus}-rank run failed (torch.distributed.run exit code {proc.returncode})")
    if line is None:
        sys.exit(f"{os.path.basename(script or __file__)}: the ranks finished without a result line")
    print(line, flush=True)


def cpu_baseline(width, height, n_frames):
    """Times the CPU path on the host cores: an even sample of n_frames animation frames at the bench size."""
    from oracle import pyoracle as po
    meshes = po.default_meshes()
    cam, light = po.default_camera(), po.default_light()
    use_ref = po.ref_available()
    if use_ref:
        soups = [po.load_obj(os.path.join(po.ASSETS, n + ".obj")) for n in ("drone", "treasure", "ground")]
        scene = po.RefScene(soups, [m.rgba for m in meshes])
        threads = po.ref().refc_max_threads()
        run = lambda f: scene.render(po.animation_transforms(f), cam, light, width, height)          # noqa: E731
    else:
        threads = po.lib().orc_max_threads()
        run = lambda f: po.render_frame(meshes, po.animation_transforms(f), cam, light, width, height)   # noqa: E731
    run(0)                                                   # warm-up
    frames = [(k * 96) // n_frames for k in range(n_frames)]
    rays, t0 = 0, time.perf_counter()
    for f in frames:
        fr = run(f)
        rays += width * height + int((fr.hit_id >= 0).sum())
    dt = time.perf_counter() - t0
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": int(threads), "cpu_model": cpu_model,
            "kind": "reference" if use_ref else "port",
            "sample": f"{n_frames} frames of the animation ({frames[0]}, {frames[1] if n_frames > 1 else frames[0]}, ... {frames[-1]}) at {width}x{height}, "
                      f"{'compiled reference functions (gcc -O3 -ffp-contract=off) + composition loop' if use_ref else 'CPU restatement'}, "
                      f"OpenMP dynamic rows, {threads} threads, {dt:.1f} s",
            "ms_per_frame": round(dt / n_frames * 1e3, 2)}


def gather_budget(width, height, frames_per_s_per_gpu, world, measured):
    """What the gather of RGB8 framebuffers to rank 0 (scene.c:186-206 consumes them there) asks of the links at this
    run's per-GPU frame rate.  Every peer has its own xGMI link to rank 0, so the gather is one point-to-point stream
    per link; rank 0 takes in the sum."""
    frame = width * height * 3
    per_peer = frames_per_s_per_gpu * frame / 1e9
    out = {"gather_bytes_per_step": frame,
           "per_gpu_frames_per_s": round(frames_per_s_per_gpu, 1),
           "per_peer_GBs_required": round(per_peer, 2),
           "xgmi_link_GBs_one_way_assumed": XGMI_LINK_GBS_ONE_WAY,
           "link_share_required": round(per_peer / XGMI_LINK_GBS_ONE_WAY, 3),
           "rank0_ingest_GBs_required": {str(n): round(per_peer * (n - 1), 1) for n in (2, 4, 8)},
           "note": ("the gathers of this run are inside the timed region: `value` already pays for them" if measured else
                    "projection from a run without peers -- unmeasured on hardware: each peer must push per_peer_GBs_required over its own "
                    "link to rank 0 (link_share_required of one direction of that link) for N GPUs to render N times this rate; rank 0's "
                    "HBM takes rank0_ingest_GBs_required in, a few percent of its 8 TB/s")}
    return out


def issue_model(slabs, tris, rays, alg_flops, kernel_seconds):
    """The reference's arithmetic priced in vector-issue cycles of gfx950 (scripts/valu_issue_bench.hip, measured: a wave64
    add / subtract / multiply occupies its SIMD for 2.1 cycles, a fused multiply-add 2.4, min / max / compare / select /
    convert and the two-wide packed forms 4.0-4.1, reciprocal and square root 8.0; 1024 SIMDs).  Per box test (aabb.c:33-56 with
    the reciprocals hoisted): 6 subtractions, 6 multiplications, 6 min/max and two three-way min/max.  Per triangle test
    (ray.c:42-67): 36 add/sub/mul, one correctly rounded division (10 instructions), 6 compares.  Per ray and mesh
    (ray.c:3-23 hoisted to a 3x4 transform, a normalisation): 30 add/sub/mul, a square root, a division.  The floor is the
    time these instructions alone would take with every lane busy and nothing else issued."""
    cost = {"add_sub_mul": 2.11, "fma": 2.44, "min_max_cmp_select": 4.03, "rcp_sqrt": 8.03}
    src = None
    try:
        t = json.load(open(ISSUE_PROFILE))["cycles_per_wave_instruction"]
        cost = {"add_sub_mul": t["v_add_f32"]["all64@8"], "fma": t["v_fma_f32"]["all64@8"],
                "min_max_cmp_select": t["v_min_f32"]["all64@8"], "rcp_sqrt": t["v_rcp_f32"]["all64@8"]}
        src = "profiles/" + os.path.basename(ISSUE_PROFILE)
    except Exception:
        pass
    div = cost["rcp_sqrt"] + 4 * cost["fma"] + 5 * cost["min_max_cmp_select"]      # v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup + a select
    box = 12 * cost["add_sub_mul"] + 8 * cost["min_max_cmp_select"]
    tri = 36 * cost["add_sub_mul"] + div + 6 * cost["min_max_cmp_select"]
    xform = 30 * cost["add_sub_mul"] + cost["rcp_sqrt"] + div
    wave_cycles = (slabs * box + tris * tri + rays * 3 * xform) / 64.0                 # 64 lanes per wave instruction
    clock_hz, simds = 2.4e9, 1024
    floor_s = wave_cycles / (simds * clock_hz)
    return {"cycles_per_wave64_instruction": cost, "source": src,
            "cycles_per_box_test_wave": round(box, 1), "cycles_per_triangle_test_wave": round(tri, 1), "cycles_per_ray_transform_wave": round(xform, 1),
            "arithmetic_floor_ms": round(floor_s * 1e3, 4),
            "ceiling_TFLOPs": round(alg_flops / floor_s / 1e12, 2),
            "frac_of_ceiling": round(floor_s / kernel_seconds, 4),
            "basis": "the reference-equivalent tests of the launch priced at the measured issue cost of their own float operations, all 64 lanes "
                     "busy, no address, stack or control instructions, 2.4 GHz: the time below which no bit-exact vector-ALU kernel can "
                     "render them; the kernel's ordered, culled traversal runs fewer tests than that count, which is how frac_of_ceiling "
                     "can approach or pass 1"}


class PatternScene:
    """--rehearse-cpu: stands in for the renderer with byte patterns, so that the multi-rank flow runs without a GPU."""

    def __init__(self):
        self.ctx = None

    @staticmethod
    def fill(frames, out):
        for k, f in enumerate(frames):
            out[k].fill_((37 * f + 11) % 251)

    def render_frames(self, frames, W, H, out, stream=None):
        self.fill(frames, out)

    def close(self):
        pass


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return spawn_ranks(a, argv)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if a.batch <= 0:
        a.batch = FRAMES_PER_LAUNCH if world == 1 else FRAMES_PER_LAUNCH // 2
    rehearse = a.rehearse_cpu
    if rehearse:
        a.backend = "gloo"
    elif not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the rendering core has no CPU path")
    if a.one_device:
        local = 0
    if rehearse:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    grouped = world > 1 or (a.collective_at_one_rank and "WORLD_SIZE" in os.environ)
    if grouped:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    def say(msg):
        if a.verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    def device_sync():
        if not rehearse:
            torch.cuda.synchronize(dev)

    from raytracer_amd import sharding
    W, H = a.width, a.height
    if rehearse:
        scene, ctx, NF, sptr, stream = PatternScene(), None, 96, None, None
    else:
        from raytracer_amd import capi, demo
        scene = demo.DemoScene(local)       # its loader's printf lines go to stderr: stdout is kept for the one JSON line
        ctx = scene.ctx
        if a.block:
            ctx.set_option("block", a.block)
        for item in a.set:
            name, _, number = item.partition("=")
            ctx.set_option(name, int(number))
        NF = demo.N_FRAMES
        # An explicit stream, made torch's current one: the kernels are launched on it, and RCCL orders
        # its gathers after the work already queued on the current stream.
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        sptr = stream.cuda_stream
    say("scene loaded")

    # frames of this rank's timed steps (independent of --warmup) and of its warm-up steps
    share = sharding.frames_for_rank(NF, rank, world) or [rank % NF]
    seq = frame_sequence(share, a.steps)
    warm = frame_sequence(share, a.steps + a.warmup)[a.steps:]

    # ---- untimed: rays and reference-equivalent work of every distinct frame of this rank
    info, own = {}, {}
    if rehearse:
        for f in set(seq):
            info[f] = {"primary_rays": W * H, "shadow_rays": 0}
    else:
        hit_id = torch.empty((H, W), dtype=torch.int32, device=dev)
        scratch = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        for f in sorted(set(seq)):
            scene.render(f, W, H, scratch, hit_id=hit_id, flags=capi.RT_FLAG_COUNT_WORK | capi.RT_FLAG_REFERENCE_ORDER, stream=sptr)
            stream.synchronize()
            info[f] = ctx.last_counters()                       # the reference's traversal: every box it would test
            assert info[f]["hit_pixels"] == int((hit_id >= 0).sum().item())
            scene.render(f, W, H, scratch, flags=capi.RT_FLAG_COUNT_WORK, stream=sptr)
            stream.synchronize()
            own[f] = ctx.last_counters()                        # the traversal the timed kernel runs (ordered, culled)
        ctx.set_option("timing", 0)
    say("work counted")

    # ---- warm-up
    # launches of equal size (to within one frame), none larger than a.batch: 20 steps are two launches of 10, not 16 + 4 --
    # a launch's last tiles keep the whole GPU waiting, and they weigh most in a small launch
    nb = (a.steps + a.batch - 1) // a.batch
    per = (a.steps + nb - 1) // nb
    sizes = [per if b < a.steps - nb * (per - 1) else per - 1 for b in range(nb)]
    bounds = [sum(sizes[:b]) for b in range(nb + 1)]
    batches = [seq[bounds[b]:bounds[b + 1]] for b in range(nb)]
    RING = 4                                                   # batches in flight: send buffers and gathers
    bufs = [torch.empty((per, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(min(nb, RING))]
    for buf in bufs:                                           # first touch of every output buffer happens here, untimed
        buf.zero_()                                            # (on the current stream = the render stream)
    # The timed region of the default run lasts some 15 ms, less than the GPU needs to reach its working clocks
    # after the synchronous frame-by-frame counting above.  So the GPU is kept busy for a fixed number of untimed
    # launches first, whatever --warmup says; the W warm-up steps proper follow.
    if not rehearse:
        for k in range(SETTLE):                                 # the launches of the timed region, in turn
            scene.render_frames(batches[k % nb], W, H, bufs[k % len(bufs)], stream=sptr)
    for k, i in enumerate(range(0, a.warmup, per)):
        scene.render_frames(warm[i:min(i + per, a.warmup)], W, H, bufs[k % len(bufs)], stream=sptr)
    if grouped:
        g = sharding.FrameGather(rank, world, collective_at_one_rank=grouped)
        g.submit(bufs[0])
        g.finish()
    device_sync()
    say("warm-up rendered and gathered")
    if grouped:
        dist.barrier()

    # ---- timed region: exactly `steps` frames, repeated
    # rank 0 drops the gathered frames (an encoder would consume them here)
    # (a rehearsal checks them instead: every frame of every rank's batches carries the byte its frame number gives it)
    verified = [0, 0]                                          # frames checked, frames right

    def check_patterns(seq_no, per_rank):
        b = seq_no % nb
        for r, t in enumerate(per_rank):
            sh = sharding.frames_for_rank(NF, r, world) or [r % NF]
            theirs = frame_sequence(sh, a.steps)[bounds[b]:bounds[b + 1]]
            for k, f in enumerate(theirs):
                verified[0] += 1
                verified[1] += int(bool((t[k] == (37 * f + 11) % 251).all()))

    gather = (sharding.FrameGather(rank, world, ring=RING - 1, consume=check_patterns if rehearse else None, collective_at_one_rank=grouped)
              if grouped else None)
    if gather is not None:
        gather.prealloc(nb, bufs[0])

    def timed_pass():
        if grouped:
            dist.barrier()
        device_sync()
        t0 = time.perf_counter()
        for b in range(nb):
            scene.render_frames(batches[b], W, H, bufs[b % len(bufs)], stream=sptr)
            if gather is not None:
                gather.submit(bufs[b % len(bufs)])
        if gather is not None:
            gather.finish()
        device_sync()
        if grouped:
            dist.barrier()
        return time.perf_counter() - t0

    times = [timed_pass()]
    repeats = a.repeats if a.repeats > 0 else min(MAX_REPEATS, max(MIN_REPEATS, int(MIN_TIMED_SECONDS / max(times[0], 1e-6)) + 1))
    if rehearse and a.repeats <= 0:
        repeats = 3
    if grouped:                                                # every rank runs the same number of repetitions: the largest wish
        want = torch.tensor([repeats], dtype=torch.int64, device=dev)
        dist.all_reduce(want, op=dist.ReduceOp.MAX)
        repeats = int(want.item())
    while len(times) < repeats:
        times.append(timed_pass())
    say(f"timed region: {repeats} repetitions, median {sorted(times)[len(times) // 2]:.4f} s")

    # Rays of the timed steps.  The kernel stores the colour only, and sends no shadow ray from a surface whose own term
    # max(0.2, n.L) is 0.2 already (scene.c:143-144 gives 0.2 blocked or not): those rays (counted by the untimed runs above,
    # which send them) are left out of `value`, which so counts the rays that were traversed.
    idle_rank = sum(info[f].get("idle_shadow_rays", 0) for f in seq)
    rays_rank = sum(info[f]["primary_rays"] + info[f]["shadow_rays"] for f in seq) - idle_rank
    if grouped:
        tmax = torch.tensor(times, dtype=torch.float64, device=dev)      # per repetition: the slowest rank
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rsum = torch.tensor([float(rays_rank)], dtype=torch.float64, device=dev)
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
        per_rep, rays_all = [float(x) for x in tmax.tolist()], float(rsum.item())
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, "cpu" if rehearse else f"cuda:{local}")
    else:
        per_rep, rays_all = list(times), float(rays_rank)
        rank_devices = ["cpu" if rehearse else f"cuda:{local}"]
    ordered = sorted(per_rep)
    dt_all = ordered[len(ordered) // 2]                              # the median repetition

    # ---- untimed: what arrives on rank 0 is what the ranks rendered.  Every rank sends the first batch of its
    # timed steps once more; rank 0 renders the same frames itself and compares the bytes.
    gather_ok = None
    if grouped:
        def first_batch(r):
            sh = sharding.frames_for_rank(NF, r, world) or [r % NF]
            fr = frame_sequence(sh, per)
            return fr
        scene.render_frames(first_batch(rank), W, H, bufs[0], stream=sptr)
        g = sharding.FrameGather(rank, world, collective_at_one_rank=grouped)
        g.submit(bufs[0])
        got = g.finish()
        device_sync()
        if rank == 0:
            mine = torch.empty_like(bufs[0])
            gather_ok = True
            for r in range(world):
                scene.render_frames(first_batch(r), W, H, mine, stream=sptr)
                device_sync()
                gather_ok = gather_ok and bool(torch.equal(mine, got[0][r].to(dev)))
        say(f"gather verified: {gather_ok}")

    # ---- kernel duration with HIP events on the launch stream (rank 0): the same launches as the
    # timed region, one at a time
    roof, single_ms, fast = None, None, None
    if rank == 0 and not rehearse:
        ctx.set_option("timing", 1)

        def work(c):
            rays = c["primary_rays"] + c["shadow_rays"]
            return (c["slab_tests"] * B_SLAB + c["tri_tests"] * B_TRI + c["hit_pixels"] * B_HIT + W * H * B_PIXEL,
                    c["slab_tests"] * F_SLAB + c["tri_tests"] * F_TRI + rays * 3 * F_XFORM)
        ms, alg_bytes, alg_flops, own_bytes, own_flops = [], [], [], [], []
        for fr in batches:
            scene.render_frames(fr, W, H, bufs[0], stream=sptr)
            ms.append(ctx.last_kernel_ms())
            alg_bytes.append(sum(work(info[f])[0] for f in fr)); alg_flops.append(sum(work(info[f])[1] for f in fr))
            own_bytes.append(sum(work(own[f])[0] for f in fr)); own_flops.append(sum(work(own[f])[1] for f in fr))
        # full-size launches only (the last launch of the region may hold fewer frames)
        full = [i for i, fr in enumerate(batches) if len(fr) == per] or list(range(len(batches)))
        ms, alg_bytes, alg_flops = [ms[i] for i in full], [alg_bytes[i] for i in full], [alg_flops[i] for i in full]
        own_bytes, own_flops = [own_bytes[i] for i in full], [own_flops[i] for i in full]
        launch_frames = float(np.mean([len(batches[i]) for i in full]))
        kern_ms = float(np.mean(ms))
        secs = kern_ms * 1e-3
        tfl = float(np.mean(alg_flops)) / secs / 1e12
        roof = {"bound": "valu", "achieved": round(tfl, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tfl / VALU_PEAK_TFLOPS, 5), "traffic": None, "hbm_frac": None,
                "kernel": "render_batch (block %d), %d frames per launch (the %d launches of that size in the region)" % (ctx.get_option("block"), per, len(full)),
                "kernel_ms": round(kern_ms, 4), "algorithmic_flops_per_launch": int(np.mean(alg_flops)),
                "basis": "FLOPs of the reference-equivalent traversal (every box and triangle test scene.c/bvh.c would run: 22 per box "
                         "test, 50 per triangle test, 45 per ray and mesh) over the kernel's mean HIP-event duration; the peak is a fused "
                         "multiply-add on every lane every two cycles, the bit-exact build has one rounding per reference operation (no FMA) "
                         "and most of its operations (min, max, compare, select) issue at half that rate: see issue_model",
                "algorithmic_bytes_per_launch": int(np.mean(alg_bytes)),
                "cache_served_GBs": round(float(np.mean(alg_bytes)) / secs / 1e9, 1),
                "kernel_counted": {"flops_per_launch": int(np.mean(own_flops)), "bytes_per_launch": int(np.mean(own_bytes)),
                                   "box_tests_per_launch": int(np.mean([sum(own[f]["slab_tests"] for f in batches[i]) for i in full])),
                                   "triangle_tests_per_launch": int(np.mean([sum(own[f]["tri_tests"] for f in batches[i]) for i in full])),
                                   "TFLOP/s": round(float(np.mean(own_flops)) / secs / 1e12, 3),
                                   "basis": "tests the ordered, distance-culled traversal of the kernel executes, counted by runs that send every shadow ray: "
                                            "the tests of the rays the timed kernel leaves out (see rays) are included"},
                "note": "the scene (150 KB of nodes, 170 KB of triangles) is cache resident: the algorithmic bytes are served by "
                        "L1/L2 (cache_served_GBs), HBM carries the framebuffers and texels only"}
        # the same launches with the fast build of the kernels (context option "fast": fused multiply-adds, approximate
        # reciprocals; not bit-exact, inside the tolerance tests/test_gpu_parity.py asserts).  Never part of `value`.
        ctx.set_option("fast", 1)
        scene.render_frames(batches[0], W, H, bufs[0], stream=sptr)      # first use of the fast build's code object: not timed
        fms = []
        for fr in batches:
            scene.render_frames(fr, W, H, bufs[0], stream=sptr)
            fms.append(ctx.last_kernel_ms())
        ctx.set_option("fast", 0)
        fms = [fms[i] for i in full]                                     # the launches `kernel_ms` above is the mean of
        fast = {"kernel_ms": round(float(np.mean(fms)), 4), "speedup_over_exact_kernel": round(kern_ms / float(np.mean(fms)), 3),
                "note": "option \"fast\" (not the default, not in `value`): fused arithmetic, and the treasure walked through a tree over "
                        "its triangles themselves (leaves of up to 8 neighbours in tight boxes) instead of the reference's leaves (the drone keeps "
                        "its tree in a sheared frame, which prunes more); hit ids equal on "
                        ">= 99.99 % of camera rays, colours within 1 on >= 99.9 % of pixels"}
        # one frame launched alone (its duration is set by its slowest tiles)
        one = []
        for f in sorted(set(seq))[:8]:
            scene.render(f, W, H, bufs[0][0], stream=sptr)
            one.append(ctx.last_kernel_ms())
        single_ms = round(float(np.mean(one)), 4)
        roof["issue_model"] = issue_model(float(np.mean([sum(info[f]["slab_tests"] for f in batches[i]) for i in full])),
                                          float(np.mean([sum(info[f]["tri_tests"] for f in batches[i]) for i in full])),
                                          float(np.mean([sum(info[f]["primary_rays"] + info[f]["shadow_rays"] for f in batches[i]) for i in full])),
                                          float(np.mean(alg_flops)), secs)
        if os.path.exists(PMC_PROFILE):
            try:        # counters of a separate rocprofv3 --pmc pass over this same command (scripts/gpu_pmc.sh), per launch
                pmc = json.load(open(PMC_PROFILE))
                hbm = pmc.get("hbm_bytes_per_launch")
                if hbm:
                    # per frame from the counter pass, times the number of frames in this run's full-size launches
                    hbm = int(hbm / float(pmc.get("frames_per_launch", 8)) * launch_frames)
                roof["traffic"] = hbm
                if hbm:
                    roof["hbm_frac"] = round(hbm / secs / 1e9 / HBM_PEAK_GBS, 5)
                roof["traffic_source"] = ("profiles/%s: FETCH_SIZE (x2 on gfx950) + WRITE_SIZE per %s launch (scaled to %.0f frames per "
                                          "launch), a separate --pmc pass of the builder's over this command's launches, not this run's"
                                          % (os.path.basename(PMC_PROFILE), pmc.get("kernel"), launch_frames))
                roof["valu_busy_estimate_pmc"] = pmc.get("valu_busy_estimate")    # [low, high]: per-class instruction counts x measured issue costs
                roof["valu_lane_utilisation_pmc"] = pmc.get("valu_lane_utilisation")
                roof["vector_memory_path_busy_pmc"] = pmc.get("ta_busy")          # TA_TA_BUSY over the 256 CUs: the next limit after issue
            except Exception:
                pass

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not rehearse:
        cpu = cpu_baseline(W, H, a.cpu_frames)

    if rank == 0:
        transport = {"nccl": "RCCL", "gloo": "gloo (host memory)"}.get(a.backend, a.backend)
        out = {
            "metric": "Mrays/s (primary+shadow), 1080p drone scene",
            "value": round(rays_all / dt_all / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt_all / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "reference scene assets (drone/treasure/ground OBJ + WebP), animation frames of raytracer.c",
            "config": {"workload": f"{W}x{H} frame, 1 primary ray/pixel + 1 shadow ray/hit pixel, 3 meshes (3625 triangles)",
                       "frames": f"step k of a rank renders entry ({FRAME_STRIDE} k) mod L of its L frames of the 96-frame animation: "
                                 "an even sample of the animation for any --steps",
                       "frames_per_rank": a.steps, "frames_per_launch": per, "launch_sizes": sizes, "untimed_settling_launches": SETTLE,
                       "sharding": "frame f on rank f mod N" if world > 1 else "single GPU",
                       "gather": f"RGB8 framebuffers to rank 0 over {transport}, {per} frames per message" if grouped else "none",
                       "world_size": dist.get_world_size() if grouped else 1,
                       "backend": (dist.get_backend() if grouped else None),
                       "rank_devices": rank_devices,
                       "gathered_frames_equal_rank0_render": gather_ok},
            "timing": {"repeats": len(per_rep), "value_from": "median repetition (each: exactly `steps` frames between barrier + synchronize, slowest rank)",
                       "timed_seconds_total": round(sum(per_rep), 4),
                       "ms_per_step_min_median_max": [round(ordered[0] / a.steps * 1e3, 4), round(dt_all / a.steps * 1e3, 4), round(ordered[-1] / a.steps * 1e3, 4)],
                       "value_min_max": [round(rays_all / ordered[-1] / 1e6, 2), round(rays_all / ordered[0] / 1e6, 2)]},
            "gather_budget": gather_budget(W, H, a.steps / dt_all, world, grouped and world > 1),
            "frames_per_s": round(a.steps * world / dt_all, 2),
            "rays_per_frame": int(rays_rank / a.steps),
            "rays": {"per_frame_in_value": int(rays_rank / a.steps),
                     "per_frame_of_the_reference": int((rays_rank + idle_rank) / a.steps),
                     "shadow_rays_not_sent_per_frame": int(idle_rank / a.steps),
                     "value_counting_the_reference_rays": round(rays_all * (1.0 + idle_rank / max(rays_rank, 1)) / dt_all / 1e6, 2),
                     "note": "rank 0's frames; the reference sends a shadow ray from every hit (scene.c:117-139); the kernel stores the colour only and "
                             "sends none from a surface whose term max(0.2, n.L) is 0.2 already -- scene.c:143-144 gives diffuse 0.2 blocked "
                             "or not, the bytes are the reference's -- and `value` does not count those rays"},
            "single_frame_kernel_ms": single_ms,
            "roofline": roof,
            "fast_build": fast,
            "cpu_baseline": cpu,
        }
        if rehearse:
            out["rehearsal"] = "cpu: no GPU, no rendering, frames are byte patterns -- value is not a measurement"
            out["config"]["rehearsal_frames_checked_on_rank0"] = verified[0]
            out["config"]["rehearsal_frames_right"] = verified[1]
            out["data"] = "byte patterns"
        print(json.dumps(out), flush=True)
    scene.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
