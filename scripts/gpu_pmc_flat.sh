#!/bin/bash
# Counter passes over scripts/flat_bench.py --profile (the one-mesh stress scene), one rocprofv3 run per pass.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_flat
rm -rf "$OUT"; mkdir -p "$OUT"
run_pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 scripts/flat_bench.py --profile --reps 3 > "$OUT/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run_pass sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU || exit 1
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
run_pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
python3 scripts/pmc_summary.py "$OUT" | grep -A12 "render_" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
