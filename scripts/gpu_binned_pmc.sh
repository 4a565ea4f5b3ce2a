set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_binned
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d gpurun_out/pmc_binned -o b -- python3 scripts/instanced_bench.py --binned 1 --no-rebuild --frames 0 --reps 2 --chunk 2 > gpurun_out/pmc_binned.json 2> gpurun_out/pmc_binned.err
rc=$?; echo "pmc exit $rc"; [ $rc -ne 0 ] && { tail -5 gpurun_out/pmc_binned.err; exit $rc; }
f=$(find gpurun_out/pmc_binned -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,v in acc.items():
    d=max(n[k],1)
    print(k, 'dispatches',d, {c: round(x/d/1e6,2) for c,x in v.items()}, 'lane util', round(v['SQ_THREAD_CYCLES_VALU']/max(v['SQ_ACTIVE_INST_VALU'],1)/64,3) if 'SQ_ACTIVE_INST_VALU' in v else '')
PY
