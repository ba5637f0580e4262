run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms')"; }
for f in 0 16 8 24 2; do
  echo "== cb_adv debug flags $f"; SSA_DEBUG_FLAGS=$f timeout -k 10 200 python -u bench.py --workload cb_adv --rows 1000000000 --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done
