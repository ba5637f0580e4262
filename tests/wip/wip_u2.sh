run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],2),'ms')"; }
echo "== U=4"; SSA_KEYED_U=4 timeout -k 10 200 python -u bench.py --workload q1 --no-cpu-baseline --steps 10 2>/dev/null | run || exit 1
for b in 2 3 4 5 6; do
  echo "== U=2 blocks/CU $b"; SSA_DEBUG_PLAN=1 SSA_KEYED_U=2 SSA_SPEC_BLOCKS_PER_CU=$b timeout -k 10 200 python -u bench.py --workload q1 --no-cpu-baseline --steps 10 2>gpurun_out/u2_$b.err | run || { tail -3 gpurun_out/u2_$b.err; exit 1; }
  grep -m2 "specialised\|keyed:" gpurun_out/u2_$b.err
done
