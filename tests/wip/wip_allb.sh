run() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms', d['config']['kernel_build_ms_before_the_timed_steps'],'ms build')"; }
for w in q1 q1_codes q6; do
  echo "== $w"; timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --steps 10 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done
echo "== cb_q20"; timeout -k 10 200 python -u bench.py --workload cb_q20 --rows 25000000 --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
for w in cb_userid cb_phrase; do
  echo "== $w"; timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done
for w in cb_q1 cb_q2 cb_q6 cb_adv; do
  echo "== $w 1g"; timeout -k 10 200 python -u bench.py --workload $w --rows 1000000000 --no-cpu-baseline --steps 5 2>gpurun_out/adv.err | run || { tail -3 gpurun_out/adv.err; exit 1; }
done
