#!/bin/bash
# Instanced GPU tests, then the stress scene with the nested loops and instance by instance, then a kernel trace of the latter.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_instanced.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/inst_tests.log 2>&1
rc=$?; tail -n 25 gpurun_out/inst_tests.log; echo "pytest exit $rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u scripts/instanced_bench.py --binned 0 --no-rebuild > gpurun_out/inst_b0.json 2> gpurun_out/inst_b0.err
rc=$?; python3 -c "import json;d=json.load(open('gpurun_out/inst_b0.json'));print('nested', d['frames'])"; [ $rc -ne 0 ] && { tail -5 gpurun_out/inst_b0.err; exit $rc; }
for c in 0 2 4 8; do
timeout -k 10 200 python -u scripts/instanced_bench.py --binned 1 --no-rebuild --chunk $c > gpurun_out/inst_c$c.json 2> gpurun_out/inst_c$c.err
rc=$?; python3 -c "import json;d=json.load(open('gpurun_out/inst_c$c.json'));print('binned chunk', $c, d['binned_path_taken'], d['binned_plan'], d['frames'])"; [ $rc -ne 0 ] && { tail -5 gpurun_out/inst_c$c.err; exit $rc; }
done
rm -rf gpurun_out/prof_binned
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_binned -o binned -- python3 scripts/instanced_bench.py --binned 1 --no-rebuild --frames 0 --reps 10 "$@" > gpurun_out/prof_binned.json 2> gpurun_out/prof_binned.err
rc=$?; echo "rocprof exit $rc"; [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_binned.err; exit $rc; }
f=$(find gpurun_out/prof_binned -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -n 12 "$f" | cut -c1-200
