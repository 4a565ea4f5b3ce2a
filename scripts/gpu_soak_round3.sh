#!/bin/bash
# The families of scripts/soak_parity.py on the build at hand, in two parts that each fit one GPU call
# (usage: gpu_soak_round3.sh a|b|c|d; c and d are the long runs); records go to gpurun_out/soak_r3_<part>.jsonl.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
PART=${1:-a}
OUT=gpurun_out/soak_r3_$PART.jsonl
: > $OUT
run() {
    timeout -k 10 ${RUN_TIMEOUT:-500} python -u scripts/soak_parity.py "$@" 2>gpurun_out/soak_r3.err | cut -c1-700 | tee -a $OUT
    rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "soak $* exit $rc"; tail -2 gpurun_out/soak_r3.err; exit $rc; }
}
if [ "$PART" = a ]; then
    run --scenes 12000 --seed 281 --wild 0.05 --all-paths
    run --instanced --scenes 4000 --seed 282 --wild 0.05
    run --instanced --mutate --scenes 2000 --seed 283 --wild 0.02
elif [ "$PART" = c ]; then
    run --scenes 100000 --seed 381 --wild 0.04 --all-paths
    run --instanced --scenes 20000 --seed 382 --wild 0.05
elif [ "$PART" = d ]; then
    run --soups --scenes 20000 --seed 384 --wild 0.02
    run --multi --scenes 20000 --seed 385 --wild 0.02
else
    run --soups --scenes 4000 --seed 284 --wild 0.02
    run --multi --scenes 4000 --seed 285 --wild 0.02
    run --rebuilds --scenes 1000 --seed 286 --wild 0.01
    run --full-size --all-paths --scenes 120 --seed 287
    run --scenes 2000 --seed 288 --wild 0.05 --spp 4
fi
exit 0
