run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms')"; }
for u in 8 16; do for b in 4 6 8; do
  echo "== cb_adv U=$u blocks $b"; SSA_KEYED_U=$u SSA_SPEC_BLOCKS_PER_CU=$b timeout -k 10 200 python -u bench.py --workload cb_adv --rows 1000000000 --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done; done
