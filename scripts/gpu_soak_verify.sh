#!/bin/bash
# The self-checking soak: random scenes rendered with RT_FLAG_VERIFY_SKIPS, tame and with special values mixed in.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
: > gpurun_out/soak_verify.jsonl
S=${SEED_BASE:-31}
for args in "--scenes ${SCENES:-4000} --seed $S" "--scenes ${SCENES:-4000} --seed $((S + 1)) --wild 0.05" "--scenes ${SCENES:-4000} --seed $((S + 2)) --wild 0.3"; do
  timeout -k 10 900 python -u scripts/soak_parity.py --verify-skips $args >> gpurun_out/soak_verify.jsonl 2> gpurun_out/soak_verify.err
  rc=$?; tail -n 1 gpurun_out/soak_verify.jsonl; [ $rc -ne 0 ] && { tail -3 gpurun_out/soak_verify.err; exit $rc; }
done
exit 0
