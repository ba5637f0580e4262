run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms')"; }
for b in 3 4 6 8; do
  echo "== cb_q20 blocks $b"; SSA_SPEC_BLOCKS_PER_CU=$b timeout -k 10 200 python -u bench.py --workload cb_q20 --rows 25000000 --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done
