#!/usr/bin/env python3
"""Condenses the counter passes of scripts/gpu_pmc.sh into one JSON file that bench.py reads for
the `traffic` and VALU figures of its roofline object.

    python scripts/pmc_to_profile.py gpurun_out/pmc_bench render_batch profiles/r2_traffic.json 16     # frames per profiled launch

HBM bytes per launch: FETCH_SIZE and WRITE_SIZE are reported in KiB and were collected in separate
passes.  On gfx950 FETCH_SIZE counts 64 B for every memory-side read request although wide requests
move 128 B, so it is doubled (an upper bound for narrow requests); WRITE_SIZE is taken as is.
How busy the vector pipes are: gfx950 has no counter of VALU-busy cycles.  SQ_ACTIVE_INST_VALU counts 1 per
vector instruction and 2 per reciprocal / square root whatever the instruction costs (measured on known
streams, scripts/valu_issue_bench.hip: profiles/r3_valu_issue_summary.json), so it is an instruction count, not
a time.  What an instruction costs its SIMD depends on its class -- add / subtract / multiply / move / integer
add / and: 2.1 cycles per wave64 instruction; fused multiply-add 2.4; min, max, compare, select, shift, convert,
three-operand integer forms and the two-wide packed float forms: 4.0-4.1; reciprocal, square root: 8.0 -- and the
per-class instruction counters (SQ_INSTS_VALU_ADD_F32 counts add, subtract and packed add; _MUL_F32 multiply and
packed multiply; _FMA_F32 fma, fmac, div_fmas; _TRANS_F32; _INT32; _CVT; the rest -- min, max, compare, select,
move, logic, shifts -- is counted by none) bound the busy time from both sides: `valu_busy_estimate` = [every
instruction of an ambiguous class at its cheapest member's cost, ... at its dearest member's cost] over the
launch's SIMD cycles (GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs; 1024 SIMDs).  An optional fifth
argument gives the two-wide packed float operations per launch as the kernel's own work counters see them, per ray
(six per box test); divided by 64 lanes and the measured lane utilisation they become wave instructions, which the
high estimate prices at the packed cost.
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
    packed = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
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
        # lanes switched on per vector instruction (SQ_THREAD_CYCLES_VALU counts 64 for a full one, SQ_ACTIVE_INST_VALU one per instruction
        # and two per transcendental)
        "valu_lane_utilisation": round(per["SQ_THREAD_CYCLES_VALU"] / (per["SQ_ACTIVE_INST_VALU"] * 4 * 16), 4),
        "salu_wave_instructions": int(per.get("SQ_INSTS_SALU", 0)),
        "vmem_read_wave_instructions": int(per.get("SQ_INSTS_VMEM_RD", 0)),
        "lds_wave_instructions": int(per.get("SQ_INSTS_LDS", 0)),
        "lds_bank_conflict_cycles": int(per.get("SQ_LDS_BANK_CONFLICT", 0)),
        "l2_hit_rate": round(per.get("TCC_HIT_sum", 0) / max(per.get("TCC_HIT_sum", 0) + per.get("TCC_MISS_sum", 0), 1), 4),
        "l1_accesses": int(per.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)),
        "l1_to_l2_read_requests": int(per.get("TCP_TCC_READ_REQ_sum", 0)),
        "waves": int(per.get("SQ_WAVES", 0)),
    }
    n = per["SQ_INSTS_VALU"]
    add, mul, fma, trans = per.get("SQ_INSTS_VALU_ADD_F32", 0), per.get("SQ_INSTS_VALU_MUL_F32", 0), per.get("SQ_INSTS_VALU_FMA_F32", 0), per.get("SQ_INSTS_VALU_TRANS_F32", 0)
    int32, cvt = per.get("SQ_INSTS_VALU_INT32", 0), per.get("SQ_INSTS_VALU_CVT", 0)
    if add + mul > 0:
        rest = max(n - add - mul - fma - trans - int32 - cvt, 0.0)
        lanes = per["SQ_THREAD_CYCLES_VALU"] / (per["SQ_ACTIVE_INST_VALU"] * 4 * 16)
        packed = min(packed / (64.0 * max(lanes, 0.05)), add + mul)
        low = 2.11 * (add + mul + int32 + rest) + 2.44 * fma + 8.03 * trans + 4.03 * cvt
        high = 2.11 * (add + mul - packed) + 4.03 * packed + 4.0 * fma + 8.03 * trans + 4.12 * (int32 + rest) + 4.03 * cvt
        res["valu_instructions_by_class"] = {"add_sub_f32": int(add), "mul_f32": int(mul), "fma_f32": int(fma), "rcp_sqrt": int(trans), "int32": int(int32),
                                             "convert": int(cvt), "counted_by_no_class_counter": int(rest), "packed_float_given": int(packed)}
        res["valu_busy_estimate"] = [round(low / (cycles * N_SIMD), 4), round(high / (cycles * N_SIMD), 4)]
        res["valu_mean_cycles_per_instruction_if_saturated"] = round(cycles * N_SIMD / n, 3)
    # the vector memory path: share of cycles the 256 texture-address units were busy, and L1 accesses per CU and cycle
    if "TA_TA_BUSY_sum" in per:
        res["ta_busy"] = round(per["TA_TA_BUSY_sum"] / (256 * cycles), 4)
    res["l1_accesses_per_cu_cycle"] = round(per.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / (256 * cycles), 4)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
