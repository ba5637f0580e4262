# Profile set "r1_c": kernel traces of the three workloads at SF100 and FETCH_SIZE passes at 128 M rows.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1c
mkdir -p $O
for w in q1 q1_codes q6; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline > $O/${w}_bench_line.json 2> $O/${w}.err
done
for w in q1 q1_codes q6; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$w -o pmc_$w -- python3 $R/bench.py --steps 3 --warmup 1 --rows 128000000 --workload $w --no-cpu-baseline > $O/pmc_${w}_bench_line.json 2> $O/pmc_${w}.err
done
find $O -name "*.csv" | head -40
