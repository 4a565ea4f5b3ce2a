#!/bin/bash
# Evidence for the stress configurations in one GPU call: the sampled tests (with the 4K / 16-sample / 2-light frame), counter passes
# over render_sampled, render_instanced, the ten-million-triangle frame and upscale_frames, then the side benches, which read them.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_sampled.py -m gpu -x -q --durations=5 --timeout 300 --timeout-method thread > gpurun_out/sampled_tests.log 2>&1
rc=$?; tail -n 12 gpurun_out/sampled_tests.log; [ $rc -ne 0 ] && exit $rc
profile() {   # tag, kernel name, frames per launch, program ("" = prof_frames.py), arguments
  tag=$1; kern=$2; fpl=$3; prog=$4; shift 4
  if [ -n "$prog" ]; then PROG=$prog PASS_TIMEOUT=240 bash scripts/gpu_pmc.sh "$tag" "$@" > "gpurun_out/pmc_$tag.log" 2>&1; else bash scripts/gpu_pmc.sh "$tag" "$@" > "gpurun_out/pmc_$tag.log" 2>&1; fi
  rc=$?; grep "pass .* rc=" "gpurun_out/pmc_$tag.log" | tr '\n' ' '; echo; [ $rc -ne 0 ] && { tail -n 5 "gpurun_out/pmc_$tag.log"; return $rc; }
  python3 scripts/pmc_to_profile.py "gpurun_out/pmc_$tag" "$kern" "gpurun_out/r3_pmc_$tag.json" "$fpl" > /dev/null || return 1
  cp "gpurun_out/r3_pmc_$tag.json" profiles/
  python3 -c "import json; d=json.load(open('gpurun_out/r3_pmc_$tag.json')); print('$tag', {k: d.get(k) for k in ('hbm_bytes_per_launch', 'valu_busy_estimate', 'valu_lane_utilisation', 'l2_hit_rate', 'ta_busy')})"
}
profile sampled render_sampled 1 "" --mode sampled --frames 0 || exit 1
profile instanced render_instanced 1 "" --mode instanced --frames 0,45 || exit 1
profile upscale upscale_frames 16 "" --mode upscale || exit 1
profile flat render_tiles 1 scripts/flat_bench.py --profile --reps 3 --frames 0 || exit 1
timeout -k 10 300 python -u scripts/sampled_bench.py > gpurun_out/r3_sampled_bench.json 2> gpurun_out/sampled_bench.err
rc=$?; cut -c1-400 gpurun_out/r3_sampled_bench.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/sampled_bench.err; exit $rc; }
timeout -k 10 300 python -u scripts/instanced_bench.py > gpurun_out/r3_instanced_bench.json 2> gpurun_out/instanced_bench.err
rc=$?; cut -c1-400 gpurun_out/r3_instanced_bench.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/instanced_bench.err; exit $rc; }
timeout -k 10 400 python -u scripts/flat_bench.py > gpurun_out/r3_flat_bench.json 2> gpurun_out/flat_bench.err
rc=$?; cut -c1-400 gpurun_out/r3_flat_bench.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/flat_bench.err; exit $rc; }
timeout -k 10 200 python -u scripts/upscale_bench.py > gpurun_out/r3_upscale_bench.json 2> gpurun_out/upscale_bench.err
rc=$?; cut -c1-400 gpurun_out/r3_upscale_bench.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/upscale_bench.err; exit $rc; }
exit 0
