for spec in "q6 640 704 768 832 896 960 1000 1024 1088 1152 1280" "q1_codes 512 640 704 736 768" "q1 384 448 480 504 512"; do set -- $spec; w=$1; shift; for g in "$@"; do
  echo "== $w grid $g"; SSA_SPEC_GRID=$g timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1),'Grows/s', round(d['roofline']['achieved']),'GB/s', round(d['roofline']['avg_kernel_ms'],2),'ms')" || exit 1
done; done
