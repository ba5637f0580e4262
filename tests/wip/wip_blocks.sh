run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms')"; }
for w in cb_q1 cb_q6 cb_adv; do for b in 3 4 5 6 8; do
  echo "== $w blocks $b"; SSA_SPEC_BLOCKS_PER_CU=$b timeout -k 10 200 python -u bench.py --workload $w --rows 1000000000 --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done; done
