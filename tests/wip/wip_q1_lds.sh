run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms')"; }
for l in "16,1" "32,1" "64,1" "128,1"; do
  echo "== q1 level-1 $l"; SSA_KEYED_LDS=$l timeout -k 10 200 python -u bench.py --workload q1 --no-cpu-baseline --steps 10 2>/dev/null | run || exit 1
done
