# ClickBench scan kernels against the number of resident workgroups per CU (SSA_SPEC_BLOCKS_PER_CU).  Run on the GPU box from the repository root.
for w in cb_q20 cb_adv cb_q1; do
  rows=1000000000; [ $w = cb_q20 ] && rows=25000000
  for pc in default 2 3 4 5 6 8; do
    if [ $pc = default ]; then unset SSA_SPEC_BLOCKS_PER_CU; else export SSA_SPEC_BLOCKS_PER_CU=$pc; fi
    timeout -k 10 120 python bench.py --workload $w --rows $rows --steps 10 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$w', '$pc', 'frac %.3f kernel %.4f ms step %.4f ms' % (r['frac'], r['avg_kernel_ms'], d['ms_per_step']))" || exit 1
  done
done
