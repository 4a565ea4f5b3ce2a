#!/bin/bash
# Times several prebuilt copies of the library one after the other on the same machine: gpurun_ab/<name>/librt_mi355.so each take
# the product library's place in turn (scripts/time_batches.py, or the program named in AB_CMD, which gets the copy's name as its
# first argument), the first one twice (start and end) to show the drift.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
cp raytracer_amd/csrc/librt_mi355.so /tmp/librt_keep.so
OUT=gpurun_out/ab_$(date +%H%M%S).jsonl
names=$(ls gpurun_ab)
first=$(echo $names | cut -d' ' -f1)
for n in $names $first; do
    cp gpurun_ab/$n/librt_mi355.so raytracer_amd/csrc/librt_mi355.so
    timeout -k 10 120 python3 ${AB_CMD:-scripts/time_batches.py} $n 7 "$@" >> $OUT 2>gpurun_out/ab_err.log || { echo "run $n failed"; tail -5 gpurun_out/ab_err.log; break; }
done
cp /tmp/librt_keep.so raytracer_amd/csrc/librt_mi355.so
cat $OUT
