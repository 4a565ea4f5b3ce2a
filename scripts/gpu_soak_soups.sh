#!/bin/bash
# Long runs of scripts/soak_parity.py --soups (random triangle soups, special values among their numbers, both builders and packers).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
: > gpurun_out/soak_soups_all.jsonl
run() {   # seed wild scenes
    timeout -k 10 1000 python -u scripts/soak_parity.py --soups --scenes $3 --seed $1 --wild $2 2>gpurun_out/soak_soups.err | cut -c1-3000 | tee -a gpurun_out/soak_soups_all.jsonl
    rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "soup soak seed $1 exit $rc"; tail -2 gpurun_out/soak_soups.err; exit $rc; }
}
run 91 0.0 8000
run 92 0.01 15000
run 93 0.05 8000
run 94 0.3 4000
