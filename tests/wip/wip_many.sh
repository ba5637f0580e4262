run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,2),'Grows/s', round(d['roofline']['avg_kernel_ms'],3),'ms kernel', round(d['ms_per_step'],3),'ms/step')"; }
for m in 256 512 1024 2048; do for w in cb_userid cb_phrase; do
  echo "== $w level-1 $m"; SSA_KEYED_MANY=$m timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done; done
