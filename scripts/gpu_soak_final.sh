#!/bin/bash
# Every family of scripts/soak_parity.py once more on the build at hand (the record of the round's last build).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
: > gpurun_out/soak_final.jsonl
run() {
    timeout -k 10 600 python -u scripts/soak_parity.py "$@" 2>gpurun_out/soak_final.err | cut -c1-700 | tee -a gpurun_out/soak_final.jsonl
    rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "soak $* exit $rc"; tail -2 gpurun_out/soak_final.err; exit $rc; }
}
run --scenes 30000 --seed 181 --wild 0.05 --all-paths
run --instanced --scenes 8000 --seed 182 --wild 0.05
run --instanced --mutate --scenes 4000 --seed 183 --wild 0.02
run --soups --scenes 6000 --seed 184 --wild 0.02
run --multi --scenes 8000 --seed 185 --wild 0.02
run --rebuilds --scenes 2000 --seed 186 --wild 0.01
run --full-size --all-paths --scenes 300 --seed 187
run --scenes 3000 --seed 188 --wild 0.05 --spp 4
