# diagnostics: LDS plan, per-phase cycles and a grid sweep of the Q1 kernels
for w in q1 q1_codes; do
  echo "== $w plan + phases"
  SSA_DEBUG_PLAN=1 SSA_PHASE_TIMING=1 timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -E "phases|specialised|lds|LDS|plan" | tail -8 || exit 1
done
for g in 448 512 576 640 704 768; do
  echo "== q1 grid $g"; SSA_SPEC_GRID=$g timeout -k 10 200 python -u bench.py --workload q1 --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],2),'ms')" || exit 1
done
