# Profile set "r1_e": kernel traces of the TPC-H workloads at SF100, the ClickBench
# group-bys whose kernels changed after the "d" set, the direct-operation tests' kernels.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1e
mkdir -p $O
for w in q1 q1_codes q6; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline > $O/${w}_bench_line.json 2> $O/${w}.err
  echo "done $w"
done
timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 2 --workload cb_adv --rows 1000000000 --no-cpu-baseline > $O/cb_adv_1g_bench_line.json 2> $O/cb_adv_1g.err
echo "done cb_adv"
for w in cb_userid cb_phrase; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline > $O/${w}_bench_line.json 2> $O/${w}.err
  echo "done $w"
done
timeout -k 10 300 python3 $R/bench.py > $O/default_bench_line.json 2> $O/default.err
echo "done default"
find $O -name "*kernel_stats.csv" | head -40
