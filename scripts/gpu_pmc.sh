#!/bin/bash
# Hardware-counter passes over scripts/prof_frames.py, one rocprofv3 run per pass (counters are
# not split over passes automatically; a pass holds at most 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 GRBM).
# usage: gpu_pmc.sh <tag> [prof_frames.py args...]        (PROG=scripts/other.py in the environment: another program, same passes)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
run_pass() {
  name=$1; shift
  timeout -s KILL ${PASS_TIMEOUT:-120} rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 ${PROG:-scripts/prof_frames.py} $ARGS > "$OUT/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
ARGS="$*"
run_pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES || exit 1
run_pass sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1
run_pass sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT || exit 1
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
run_pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || exit 1
run_pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
python3 scripts/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
