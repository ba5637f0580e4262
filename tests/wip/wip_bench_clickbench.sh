# ClickBench scan-aggregate workloads, one bench line each (gpurun_out/cb_<name>.json)
for w in cb_q1 cb_q2 cb_q3 cb_q6 cb_adv cb_userid cb_phrase; do
  echo "== $w"
  timeout -k 10 280 python -u bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/$w.json 2> gpurun_out/$w.err || { tail -5 gpurun_out/$w.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/$w.json').read().strip().splitlines()[-1]); print(round(d['value']/1e9,2),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms kernel', round(d['ms_per_step'],3), 'ms/step', d['config']['groups'],'groups', d['roofline']['kernel'], d['roofline']['launches'])"
done
