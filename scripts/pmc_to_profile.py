#!/usr/bin/env python3
"""Condenses the counter passes of scripts/gpu_pmc.sh into one JSON file that bench.py reads for
the `traffic` and VALU figures of its roofline object.

    python scripts/pmc_to_profile.py gpurun_out/pmc_bench render_batch profiles/r2_traffic.json 16     # frames per profiled launch

HBM bytes per launch: FETCH_SIZE and WRITE_SIZE are reported in KiB and were collected in separate
passes.  On gfx950 FETCH_SIZE counts 64 B for every memory-side read request although wide requests
move 128 B, so it is doubled (an upper bound for narrow requests); WRITE_SIZE is taken as is.
VALU issue utilisation: SQ_ACTIVE_INST_VALU counts in units of four cycles (one wave64 instruction
occupies a 16-lane SIMD for four cycles); the chip has 1024 SIMDs and GRBM_GUI_ACTIVE sums the busy
cycles of its 8 XCDs.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_XCD, N_SIMD = 8, 1024


def main():
    root, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
    frames_per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    tot = defaultdict(float)
    disp = defaultdict(set)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel not in row.get("Kernel_Name", ""):
                    continue
                c = row["Counter_Name"]
                tot[c] += float(row["Counter_Value"] or 0)
                disp[c].add(row["Dispatch_Id"])
    per = {c: tot[c] / max(len(disp[c]), 1) for c in tot}
    cycles = per["GRBM_GUI_ACTIVE"] / N_XCD
    fetch_raw = per["FETCH_SIZE"] * 1024.0
    write = per["WRITE_SIZE"] * 1024.0
    res = {
        "kernel": kernel,
        "source": os.path.basename(os.path.normpath(root)),
        "launches_profiled": max(len(v) for v in disp.values()),
        "frames_per_launch": frames_per_launch,
        "hbm_fetch_bytes_reported": int(fetch_raw),
        "hbm_fetch_bytes_corrected": int(fetch_raw * 2),
        "hbm_write_bytes": int(write),
        "hbm_bytes_per_launch": int(fetch_raw * 2 + write),
        "gpu_cycles_per_launch": int(cycles),
        "valu_wave_instructions": int(per["SQ_INSTS_VALU"]),
        # four cycles per wave64 instruction is the full-width cost; the ratio can pass 1 (instructions whose upper lanes are all
        # off cost less), which then simply means the vector pipes never idle
        "valu_issue_utilisation": round(min(1.0, per["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * N_SIMD)), 4),
        "valu_issue_ratio_at_4_cycles_per_instruction": round(per["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * N_SIMD), 4),
        "valu_lane_utilisation": round(per["SQ_THREAD_CYCLES_VALU"] / (per["SQ_ACTIVE_INST_VALU"] * 4 * 16), 4),
        "salu_wave_instructions": int(per["SQ_INSTS_SALU"]),
        "vmem_read_wave_instructions": int(per["SQ_INSTS_VMEM_RD"]),
        "lds_wave_instructions": int(per["SQ_INSTS_LDS"]),
        "lds_bank_conflict_cycles": int(per["SQ_LDS_BANK_CONFLICT"]),
        "l2_hit_rate": round(per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per["TCC_MISS_sum"], 1), 4),
        "l1_accesses": int(per["TCP_TOTAL_CACHE_ACCESSES_sum"]),
        "l1_to_l2_read_requests": int(per["TCP_TCC_READ_REQ_sum"]),
        "waves": int(per["SQ_WAVES"]),
    }
    # the vector memory path: share of cycles the 256 texture-address units were busy, and L1 accesses per CU and cycle
    if "TA_TA_BUSY_sum" in per:
        res["ta_busy"] = round(per["TA_TA_BUSY_sum"] / (256 * cycles), 4)
    res["l1_accesses_per_cu_cycle"] = round(per["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * cycles), 4)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
