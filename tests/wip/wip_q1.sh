run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms')"; }
for w in q1 q1_codes q1; do
  echo "== $w"; timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --steps 10 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done
