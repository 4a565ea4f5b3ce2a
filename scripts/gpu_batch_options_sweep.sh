cd "${GRAFT_REPO_ROOT:-.}"
for c in 0 256 512 2048 4096 8192; do
  v=$(timeout -k 10 120 python -u bench.py --no-cpu-baseline --set batch_chunk=$c 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])")
  echo "chunk $c: $v"
done
for r in 0 1; do
  v=$(timeout -k 10 120 python -u bench.py --no-cpu-baseline --set batch_rows=$r 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])")
  echo "batch_rows $r: $v"
done
v=$(timeout -k 10 120 python -u bench.py --no-cpu-baseline --set xcd_swizzle=0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])")
echo "xcd_swizzle 0: $v"
v=$(timeout -k 10 120 python -u bench.py --no-cpu-baseline --block 256 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])")
echo "block 256: $v"
