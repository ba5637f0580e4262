for w in cb_q1 cb_q2 cb_q3 cb_q6; do
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --workload $w --rows 1000000000 --no-cpu-baseline > gpurun_out/${w}_1g.json 2> gpurun_out/${w}_1g.err || { tail -5 gpurun_out/${w}_1g.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/${w}_1g.json').read().strip().splitlines()[-1]); print('$w', round(d['value']/1e9,2),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],3),'ms kernel', round(d['ms_per_step'],3), 'ms/step')"
done
