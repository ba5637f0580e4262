# Profile set "r1_d": kernel traces of the TPC-H workloads at SF100, FETCH_SIZE passes at 128 M rows, ClickBench
# scan-aggregates at 1 G rows (10 x hits) and group-bys at the hits size, the filter algebra kernels.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1d
mkdir -p $O
for w in q1 q1_codes q6; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline > $O/${w}_bench_line.json 2> $O/${w}.err
  echo "done $w"
done
for w in q1; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$w -o pmc_$w -- python3 $R/bench.py --steps 3 --warmup 1 --rows 128000000 --workload $w --no-cpu-baseline > $O/pmc_${w}_bench_line.json 2> $O/pmc_${w}.err
  echo "done pmc $w"
done
for w in cb_q1 cb_q2 cb_q3 cb_q6 cb_adv; do
  timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 2 --workload $w --rows 1000000000 --no-cpu-baseline > $O/${w}_1g_bench_line.json 2> $O/${w}_1g.err
  echo "done $w"
done
for w in cb_userid cb_phrase; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline > $O/${w}_bench_line.json 2> $O/${w}.err
  echo "done $w"
done
find $O -name "*stats.csv" | head -40
