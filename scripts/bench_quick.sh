#!/bin/bash
# One short bench line per workload (kernel fraction of the roofline, kernel and step time).  Run on the GPU box from the repository root:
#   bash scripts/bench_quick.sh q1 q6 cb_q1 cb_adv cb_q20
for w in "$@"; do
  rows=""
  case $w in cb_q1|cb_adv|cb_q2|cb_q3|cb_q6) rows="--rows 1000000000";; cb_q20) rows="--rows 25000000";; esac
  timeout -k 10 300 python bench.py --workload $w $rows --steps 10 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$w', 'frac %.3f kernel %.4f ms step %.4f ms %.1f G rows/s' % (r['frac'], r['avg_kernel_ms'], d['ms_per_step'], d['value'] / 1e9), r['kernel'])" || exit 1
done
