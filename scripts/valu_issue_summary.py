#!/usr/bin/env python3
"""Condenses the runs of scripts/valu_issue_bench into one JSON file.

    python scripts/valu_issue_summary.py <plain run .jsonl> <counter pass dir> <counter pass .jsonl> <out.json>

Per instruction class: cycles per wave-instruction on a SIMD at eight waves and at one wave per SIMD for the full
EXEC mask, the same for partial masks, and what the SQ counters report for the same launches -- in particular the
value of SQ_ACTIVE_INST_VALU per wave-instruction, which is what turns that counter into a utilisation.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    plain, pmc_dir, pmc_lines, out = sys.argv[1:5]
    rows = [json.loads(l) for l in open(plain) if l.startswith("{")]
    table = defaultdict(dict)
    for r in rows:
        table[r["op"]][f'{r["exec"]}@{r["waves_per_simd"]}'] = round(r["cycles_per_wave_instruction_on_the_simd"], 3)
    clocks = sorted(r["shader_clock_ghz"] for r in rows if r["shader_clock_ghz"] > 0)
    # the counter pass launches its configurations in the order it prints them
    order = [json.loads(l) for l in open(pmc_lines) if l.startswith("{")]
    per_dispatch = defaultdict(dict)
    for path in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if "issue_kernel" not in row.get("Kernel_Name", ""):
                    continue
                per_dispatch[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"] or 0)
    ids = sorted(per_dispatch)
    ids = ids[len(ids) - len(order):]                 # the warm-up launches come first
    counters = []
    for cfg, did in zip(order, ids):
        c = per_dispatch[did]
        waves = 256 * 4 * cfg["waves_per_simd"]
        insts = cfg["wave_instructions_per_wave"] * waves
        busy_cycles_per_simd = c.get("GRBM_GUI_ACTIVE", 0) / 8
        counters.append({
            "op": cfg["op"], "exec": cfg["exec"], "waves_per_simd": cfg["waves_per_simd"],
            "cycles_per_wave_instruction_shader_clock": round(cfg["cycles_per_wave_instruction_on_the_simd"], 3),
            "SQ_INSTS_VALU_per_issued": round(c.get("SQ_INSTS_VALU", 0) / insts, 4),
            "SQ_ACTIVE_INST_VALU_per_wave_instruction": round(c.get("SQ_ACTIVE_INST_VALU", 0) / insts, 4),
            "SQ_THREAD_CYCLES_VALU_per_wave_instruction": round(c.get("SQ_THREAD_CYCLES_VALU", 0) / insts, 3),
            "SQ_BUSY_CYCLES_per_GRBM_GUI_ACTIVE": round(c.get("SQ_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 0), 1), 4),
            "SQ_WAVE_CYCLES_per_wave": round(c.get("SQ_WAVE_CYCLES", 0) / waves, 1),
            "valu_ratio_at_4_cycles": round(c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(busy_cycles_per_simd * 1024, 1), 4),
        })
    # second pass, if present: what the per-class instruction counters count for each stream (full EXEC mask only)
    pmc2 = pmc_dir.rstrip("/") + "2"
    classes = {}
    if os.path.isdir(pmc2):
        per2 = defaultdict(dict)
        for path in glob.glob(os.path.join(pmc2, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if "issue_kernel" in row.get("Kernel_Name", ""):
                        per2[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"] or 0)
        ids2 = sorted(per2)[-len(order):]
        for cfg, did in zip(order, ids2):
            if cfg["exec"] != "all64":
                continue
            insts = cfg["wave_instructions_per_wave"] * 256 * 4 * cfg["waves_per_simd"]
            classes[cfg["op"]] = {k.replace("SQ_INSTS_VALU_", "").lower() or "all": round(v / insts, 3) for k, v in per2[did].items() if k != "SQ_INSTS_VALU"}
    res = {"per_class_counters_per_wave_instruction": classes, "shader_clock_ghz_median": clocks[len(clocks) // 2] if clocks else None,
           "cycles_per_wave_instruction": table, "counters": counters}
    json.dump(res, open(out, "w"), indent=1)
    for op, t in table.items():
        print(f"{op:24s} " + "  ".join(f"{k}={v}" for k, v in sorted(t.items())))
    for c in counters:
        print(json.dumps(c))


if __name__ == "__main__":
    main()
