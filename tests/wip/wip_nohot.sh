run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms')"; }
for w in cb_userid cb_phrase q1_codes; do for f in 0 8; do
  echo "== $w debug flags $f"; SSA_DEBUG_FLAGS=$f timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done; done
