#!/bin/bash
# developer loop for the HBM band factorisation: unit check, config-5 timings, single-image kernel timeline
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/chain
mkdir -p $O
timeout -k 10 200 tools/_bin/lu_unit > $O/lu_unit.log 2>&1; grep Chol $O/lu_unit.log
grep -q "all ok" $O/lu_unit.log || exit 1
timeout -k 10 300 python3 tools/eval_cfg5.py 8 3 400 2>&1 | grep -v amdgpu | tee $O/eval8.log || exit 1
timeout -k 10 200 python3 tools/eval_cfg5.py 1 2 400 2>&1 | grep -v amdgpu | tee $O/eval1.log || exit 1
for n in ${1:-1}; do
rm -rf $O/kt$n
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$n -- python3 tools/eval_cfg5.py $n 1 400 > $O/kt$n.log 2>&1 || exit 1
python3 tools/refresh_profiles.py aggregate $O/kt$n && head -8 $(find $O/kt$n -name timeline_summary.csv)
done
