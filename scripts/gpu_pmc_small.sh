#!/bin/bash
# Two counter passes (instruction mix and lane utilisation) over scripts/prof_frames.py: usage gpu_pmc_small.sh <tag> [prof_frames.py args...]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmcs_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="$*"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq1" -o p -- python3 scripts/prof_frames.py $ARGS > "$OUT/sq1.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_BRANCH --output-format csv -d "$OUT/sq2" -o p -- python3 scripts/prof_frames.py $ARGS > "$OUT/sq2.log" 2>&1 || exit 1
python3 scripts/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "render_|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_ACTIVE_INST_VALU|SQ_THREAD_CYCLES_VALU|GRBM_GUI_ACTIVE|SQ_INSTS_VMEM_RD|SQ_INSTS_LDS|SQ_INSTS_BRANCH|SQ_WAVES" "$OUT/summary.txt"
