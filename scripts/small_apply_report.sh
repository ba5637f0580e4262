#!/bin/bash
# Applies of shard-sized batches: wall time per apply, the launches of one apply (rocprofv3 kernel trace), the host's
# phases, and the phases of one SF100-sized step.  Output under gpurun_out/small_report/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/small_report
mkdir -p $O
cd $R
for w in q6 q1 q1_codes; do
    timeout -k 10 120 python3 scripts/small_apply.py $w 1048576 300 2>&1 | grep -v amdgpu.ids | tee -a $O/times.txt || exit 1
done
SSA_HOST_PHASES=1 timeout -k 10 120 python3 scripts/small_apply.py q1 1048576 3 2>&1 | grep "^apply" | tail -1 | tee $O/host_phases_q1.txt
SSA_HOST_PHASES=1 timeout -k 10 120 python3 scripts/small_apply.py q6 1048576 3 2>&1 | grep "^apply" | tail -1 | tee $O/host_phases_q6.txt
bash scripts/trace_small.sh q6 2>&1 | grep -v amdgpu.ids > $O/launches_q6.txt || exit 1
bash scripts/trace_small.sh q1 2>&1 | grep -v amdgpu.ids > $O/launches_q1.txt || exit 1
bash scripts/trace_slices.sh 2>&1 | grep -v amdgpu.ids > $O/launches_q1_slices.txt || exit 1
PYTHONPATH=$R timeout -k 10 300 python3 scripts/step_phases.py 2>&1 | grep -v amdgpu.ids > $O/step_phases.txt || exit 1
tail -3 $O/launches_q6.txt $O/step_phases.txt
