#!/bin/bash
# First GPU pass of the round: the entry point that allocates nothing (alone, in a fresh process), the whole GPU
# suite, smoke, the bench as the driver runs it and with other step counts, kernel trace, counter passes.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 240 python -u -m pytest tests/test_zz_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/noalloc_alone.log 2>&1
rc=$?; tail -n 8 gpurun_out/noalloc_alone.log; echo "noalloc alone exit $rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 12 gpurun_out/pytest_gpu.log; echo "pytest exit $rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 180 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 2 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err
rc=$?; cat gpurun_out/bench_20.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/bench_20.err; exit $rc; }
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_96.json 2> gpurun_out/bench_96.err
rc=$?; cat gpurun_out/bench_96.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/bench_96.err; exit $rc; }
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
rc=$?; echo "rocprof exit $rc"; [ $rc -ne 0 ] && exit $rc
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -n 8 "$f"
bash scripts/gpu_pmc.sh bench --mode bench --repeat 1 > gpurun_out/pmc_bench.log 2>&1
rc=$?; tail -n 30 gpurun_out/pmc_bench.log; [ $rc -ne 0 ] && exit $rc
python3 scripts/pmc_to_profile.py gpurun_out/pmc_bench render_batch gpurun_out/r2_traffic.json 16 > /dev/null
cat gpurun_out/r2_traffic.json
