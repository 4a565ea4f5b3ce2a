#!/bin/bash
# Long runs of scripts/soak_parity.py --multi (scenes of one to six random soups through the plain, batched and instanced paths).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
: > gpurun_out/soak_multi_all.jsonl
run() {   # seed wild scenes
    timeout -k 10 1000 python -u scripts/soak_parity.py --multi --scenes $3 --seed $1 --wild $2 2>gpurun_out/soak_multi.err | cut -c1-4000 | tee -a gpurun_out/soak_multi_all.jsonl
    rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "multi soak seed $1 exit $rc"; tail -2 gpurun_out/soak_multi.err; exit $rc; }
}
run 111 0.0 10000
run 112 0.01 15000
run 113 0.05 10000
run 114 0.3 5000
