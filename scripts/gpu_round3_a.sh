#!/bin/bash
# The measurement pass in one GPU call: smoke, the bench as the driver runs it and with its defaults, the rocprofv3
# kernel trace of the same command, the counter passes over the bench's own launches, the bench through RCCL with one rank.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 180 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 2 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
# counter passes over the bench's own launches first: the bench lines below quote them (profiles/r3_traffic.json)
bash scripts/gpu_pmc.sh bench --mode bench --repeat 1 > gpurun_out/pmc_bench.log 2>&1
rc=$?; tail -n 3 gpurun_out/pmc_bench.log; [ $rc -ne 0 ] && exit $rc
cp gpurun_out/pmc_bench/summary.txt gpurun_out/r3_pmc_render_batch_summary.txt
PK=$(timeout -k 10 120 python3 scripts/prof_frames.py --mode bench --count-packed 2>/dev/null | tail -n 1)
python3 scripts/pmc_to_profile.py gpurun_out/pmc_bench render_batch gpurun_out/r3_traffic.json 16 "${PK:-0}" > /dev/null
cp gpurun_out/r3_traffic.json profiles/r3_traffic.json
cat gpurun_out/r3_traffic.json
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err
rc=$?; cat gpurun_out/bench_20.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/bench_20.err; exit $rc; }
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_96.json 2> gpurun_out/bench_96.err
rc=$?; cat gpurun_out/bench_96.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/bench_96.err; exit $rc; }
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
rc=$?; echo "rocprof exit $rc"; [ $rc -ne 0 ] && exit $rc
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { head -n 8 "$f"; cp "$f" gpurun_out/r3_bench_kernel_stats.csv; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29731 bench.py --collective-at-one-rank --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_rccl_one_rank.json 2> gpurun_out/bench_rccl.err
rc=$?; grep '"metric"' gpurun_out/bench_rccl_one_rank.json | cut -c1-600; [ $rc -ne 0 ] && { tail -5 gpurun_out/bench_rccl.err; exit $rc; }
exit 0
