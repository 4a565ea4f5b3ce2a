#!/bin/bash
# Runs on the GPU box: GPU tests, smoke, a short bench and a rocprofv3 kernel trace of the bench.
# Every GPU step has its own time limit; a crash, hang or fault ends the script.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-96}

timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 25 gpurun_out/pytest_gpu.log
echo "pytest exit $rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi

timeout -k 10 180 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log
rc2=${PIPESTATUS[0]}
if [ $rc2 -ne 0 ]; then echo "smoke exit $rc2"; exit $rc2; fi

timeout -k 10 400 python -u bench.py --steps $STEPS --warmup 8 > gpurun_out/bench.json 2> gpurun_out/bench.err
rc3=$?
cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
if [ $rc3 -ne 0 ]; then echo "bench exit $rc3"; exit $rc3; fi

rm -rf gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps $STEPS --warmup 8 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
rc4=$?
echo "rocprof exit $rc4"
find gpurun_out/prof -name '*kernel_stats*' | head -3
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -n 12 "$f"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_upscale -o upscale -- python3 scripts/upscale_bench.py > gpurun_out/upscale_bench.json 2> gpurun_out/upscale.err
rc5=$?
echo "upscale rocprof exit $rc5"
cat gpurun_out/upscale_bench.json | tail -n 1
f=$(find gpurun_out/prof_upscale -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -n 4 "$f"
exit $rc
