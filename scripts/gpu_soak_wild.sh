#!/bin/bash
# Long random runs of the parity check with special values mixed in (scripts/soak_parity.py --wild), plain and instanced frames.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
: > gpurun_out/soak_wild_all.jsonl
run() {   # mode seed wild scenes
    timeout -k 10 1000 python -u scripts/soak_parity.py $1 --scenes $4 --seed $2 --wild $3 2>gpurun_out/soak_wild.err | cut -c1-6000 | tee -a gpurun_out/soak_wild_all.jsonl | cut -c1-400
    rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "soak $1 seed $2 exit $rc"; tail -2 gpurun_out/soak_wild.err; exit $rc; }
}
run --instanced 62 0.15 8000
run --instanced 63 0.05 12000
run --instanced 64 0.3 6000
run --all-paths 71 0.05 20000
run --all-paths 72 0.2 20000
