"""What the side benches (sampled, instanced, one-mesh stress scene) say about their kernels beyond a rate: the kernel's own
work against the vector peak and against the issue-cycle floor of that work (bench.issue_model), the counter profile of a
separate rocprofv3 pass if one is stored under profiles/, and a CPU baseline."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def roofline(kernel, counters, transforms, kernel_ms, profile=None, frames_profiled_per_launch=1):
    """counters: rt_last_counters of a work-counting launch of the same frame (the kernel's own box and triangle tests);
    transforms: ray transforms into a mesh's space that the frame needs (rays x meshes for the plain loop)."""
    secs = kernel_ms * 1e-3
    flops = counters["slab_tests"] * bench.F_SLAB + counters["tri_tests"] * bench.F_TRI + transforms * bench.F_XFORM
    nbytes = counters["slab_tests"] * bench.B_SLAB + counters["tri_tests"] * bench.B_TRI + counters["hit_pixels"] * bench.B_HIT
    rays = counters["primary_rays"] + counters["shadow_rays"]
    out = {"bound": "valu", "achieved": round(flops / secs / 1e12, 3), "peak": bench.VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(flops / secs / 1e12 / bench.VALU_PEAK_TFLOPS, 5), "traffic": None,
           "kernel": kernel, "kernel_ms": round(kernel_ms, 4),
           "basis": "the box and triangle tests the kernel itself counts (22 and 50 FLOP) plus 45 FLOP per ray transform, over the HIP-event "
                    "time of the uncounted launch",
           "kernel_counted": {"box_tests": counters["slab_tests"], "triangle_tests": counters["tri_tests"], "rays": rays,
                              "box_tests_per_ray": round(counters["slab_tests"] / max(rays, 1), 2),
                              "triangle_tests_per_ray": round(counters["tri_tests"] / max(rays, 1), 2), "ray_transforms": int(transforms)},
           "cache_served_GBs": round(nbytes / secs / 1e9, 1)}
    model = bench.issue_model(float(counters["slab_tests"]), float(counters["tri_tests"]), transforms / 3.0, float(flops), secs)
    out["issue_model"] = {k: model[k] for k in ("arithmetic_floor_ms", "ceiling_TFLOPs", "frac_of_ceiling", "source")}
    if profile:
        path = os.path.join(REPO, "profiles", profile)
        if os.path.exists(path):
            pmc = json.load(open(path))
            hbm = pmc.get("hbm_bytes_per_launch")
            out["traffic"] = int(hbm / max(frames_profiled_per_launch, 1)) if hbm else None
            if hbm:
                out["hbm_frac"] = round(out["traffic"] / secs / 1e9 / bench.HBM_PEAK_GBS, 5)
            out["traffic_source"] = f"profiles/{profile}: FETCH_SIZE (x2 on gfx950) + WRITE_SIZE per launch, a separate --pmc pass over the same frame"
            out["valu_busy_estimate_pmc"] = pmc.get("valu_busy_estimate")
            out["valu_lane_utilisation_pmc"] = pmc.get("valu_lane_utilisation")
            out["l2_hit_rate_pmc"] = pmc.get("l2_hit_rate")
    return out


def cpu_baseline(run, what):
    """run() renders a bounded sample with the CPU oracle and returns (rays, threads); the rate it reaches is the baseline."""
    run()                                                    # first touch
    t0 = time.perf_counter()
    n = 0
    rays = threads = 0
    while time.perf_counter() - t0 < 5.0 and n < 20:
        r, threads = run()
        rays += r
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": int(threads), "kind": "port",
            "sample": f"{what}; {n} passes, {dt:.1f} s: the CPU restatement's rate on this sample stands for the frame (scaled, not rendered in full)"}
