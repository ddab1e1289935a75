#!/bin/bash
# profile recipe (run on the GPU box through gpurun); outputs under gpurun_out/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python3 bench.py --steps 10 --warmup 3 > $OUT/bench1.log 2>&1 || { tail -20 $OUT/bench1.log; exit 1; }
cat $OUT/bench1.log | tail -1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_kt.log 2>&1 || { tail -20 $OUT/prof_kt.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/prof_pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $OUT/prof_pmc1.log 2>&1 || { tail -20 $OUT/prof_pmc1.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $OUT/prof_pmc2.log 2>&1 || { tail -20 $OUT/prof_pmc2.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_pmc3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $OUT/prof_pmc3.log 2>&1 || { tail -20 $OUT/prof_pmc3.log; exit 1; }
find $OUT -name "*.csv" | head -20
du -sh $OUT
# full evaluate (PDHG + loss + adjoint gradient by block cyclic reduction), kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eval -- python3 tools/eval_once.py > $OUT/prof_eval.log 2>&1 || { tail -20 $OUT/prof_eval.log; exit 1; }
grep "^alpha" $OUT/prof_eval.log
