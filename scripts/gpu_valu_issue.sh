#!/bin/bash
# The vector-issue microbenchmark on the GPU: plain run (cycles per wave-instruction from the shader clock), then one
# counter pass over its short form, so that SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES can be read against known streams.
# Build scripts/valu_issue_bench beforehand (make tools).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 ./scripts/valu_issue_bench > gpurun_out/valu_issue.jsonl 2> gpurun_out/valu_issue.err
rc=$?; echo "valu_issue_bench exit $rc ($(wc -l < gpurun_out/valu_issue.jsonl) lines)"; [ $rc -ne 0 ] && { tail -5 gpurun_out/valu_issue.err; exit $rc; }
rm -rf gpurun_out/valu_issue_pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES \
    --output-format csv -d gpurun_out/valu_issue_pmc -o p -- ./scripts/valu_issue_bench --pmc > gpurun_out/valu_issue_pmc.jsonl 2> gpurun_out/valu_issue_pmc.err
rc=$?; echo "counter pass exit $rc"; [ $rc -ne 0 ] && { tail -5 gpurun_out/valu_issue_pmc.err; exit $rc; }
# what the per-class instruction counters count, on the same streams
rm -rf gpurun_out/valu_issue_pmc2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 \
    --output-format csv -d gpurun_out/valu_issue_pmc2 -o p -- ./scripts/valu_issue_bench --pmc > gpurun_out/valu_issue_pmc2.jsonl 2> gpurun_out/valu_issue_pmc2.err
rc=$?; echo "class counter pass exit $rc"; [ $rc -ne 0 ] && { tail -5 gpurun_out/valu_issue_pmc2.err; exit $rc; }
python3 scripts/valu_issue_summary.py gpurun_out/valu_issue.jsonl gpurun_out/valu_issue_pmc gpurun_out/valu_issue_pmc.jsonl gpurun_out/valu_issue_summary.json
