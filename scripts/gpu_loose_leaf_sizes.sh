#!/bin/bash
# The fast build's tree over the triangles themselves with leaves of different sizes (RT_LOOSE_LEAF_TRIS, read by the packer
# for experiments): time of the batched launches and distance from the exact image (scripts/fast_mode_check.py), then the
# instanced scene with the library's own leaf size.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
OUT=gpurun_out/loose_leaf_sizes.jsonl; : > $OUT
for n in 4 6 8 12 16; do
    echo "{\"leaf_tris\": $n}" >> $OUT
    RT_LOOSE_LEAF_TRIS=$n timeout -k 10 200 python3 scripts/fast_mode_check.py >> $OUT 2>gpurun_out/loose_err.log || { echo "leaf size $n failed"; tail -5 gpurun_out/loose_err.log; exit 1; }
done
timeout -k 10 300 python3 scripts/instanced_bench.py --no-cpu-baseline > gpurun_out/instanced_fast.json 2>gpurun_out/instanced_fast.err || { echo "instanced failed"; tail -5 gpurun_out/instanced_fast.err; exit 1; }
RT_LOOSE_LEAF_TRIS=8 timeout -k 10 300 python3 scripts/instanced_bench.py --no-cpu-baseline > gpurun_out/instanced_fast8.json 2>gpurun_out/instanced_fast.err || { echo "instanced failed"; tail -5 gpurun_out/instanced_fast.err; exit 1; }
python3 - <<'PY'
import json
for l in open("gpurun_out/loose_leaf_sizes.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k != "frames"}, [ (f["ids_differing"], f["rgb_within_1"]) for f in d.get("frames", [])])
for n in ("instanced_fast", "instanced_fast8"):
    d = json.load(open(f"gpurun_out/{n}.json"))
    print(n, d["value"], [(r["kernel_ms"], r["fast_build"]) for r in d["frames"]])
PY
