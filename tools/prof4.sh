#!/bin/bash
# Profile recipe of round 4 (run on the GPU box through gpurun; outputs under gpurun_out/prof4/).
# Same rules as tools/prof2.sh / prof3.sh: every rocprofv3 --pmc pass holds counters of ONE budget class (SQ <= 8 per
# pass, FETCH_SIZE and WRITE_SIZE in passes of their own), the program comes directly after `--`, no trace domain is
# combined with --pmc.  New this round: the single-image plan (the reference's default num_samples = 1) and the two
# sum-of-regularisers PDHG kernels get counter passes (VERDICT r3 missing #4).
# usage: tools/prof4.sh [part ...]   parts: bench single cfg5 sumregs eval128 evalcfg5 nd   (default: bench single cfg5 sumregs)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof4
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
PARTS=${@:-bench single cfg5 sumregs}
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
SQ2="SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
kt()  { local tag=$1; shift; echo "== kernel trace $tag"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- "$@" > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; exit 1; }; python3 tools/refresh_profiles.py aggregate $OUT/$tag; }
pmc() { local tag=$1; local ctr=$2; shift 2; echo "== pmc $tag: $ctr"; rocprofv3 --pmc $ctr --output-format csv -d $OUT/$tag -- "$@" > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; exit 1; }; python3 tools/refresh_profiles.py aggregate $OUT/$tag; }
for part in $PARTS; do
case $part in
bench)
  python3 bench.py --steps 10 --warmup 3 > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
  tail -1 $OUT/bench.log | cut -c1-300
  kt kt_bench python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
  # counters of the headline launch (eager launches under the profiler: the whole 490-tile grid per dispatch)
  B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-graph"
  pmc pmc_bench_sq1 "$SQ1" $B
  pmc pmc_bench_fetch "FETCH_SIZE" $B
  pmc pmc_bench_write "WRITE_SIZE" $B
  ;;
single)
  # the reference's default workload: ONE 128^2 image (num_samples = 1, /root/reference/src/BPLDenoising.jl:313)
  S="python3 bench.py --images 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
  $S > $OUT/bench_single.log 2>&1 || { tail -20 $OUT/bench_single.log; exit 1; }
  tail -1 $OUT/bench_single.log | cut -c1-300
  kt kt_single $S
  S1="python3 bench.py --images 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-graph"
  pmc pmc_single_sq1 "$SQ1" $S1
  pmc pmc_single_fetch "FETCH_SIZE" $S1
  pmc pmc_single_write "WRITE_SIZE" $S1
  ;;
cfg5)
  # config 5's per-GPU share through the PDHG kernel of large images (pdhg_rows_kernel): line, trace, counters
  C="python3 bench.py --images 8 --size 1024 --alpha-map --iters 400 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
  $C > $OUT/bench_cfg5.log 2>&1 || { tail -20 $OUT/bench_cfg5.log; exit 1; }
  tail -1 $OUT/bench_cfg5.log | cut -c1-400
  kt kt_cfg5 $C
  pmc pmc_cfg5_sq1 "$SQ1" $C --no-graph
  pmc pmc_cfg5_sq2 "$SQ2" $C --no-graph
  pmc pmc_cfg5_fetch "FETCH_SIZE" $C --no-graph
  pmc pmc_cfg5_write "WRITE_SIZE" $C --no-graph
  ;;
sumregs)
  python3 tools/gpu_sumregs_time.py > $OUT/sumregs_time.log 2>&1 || { tail -20 $OUT/sumregs_time.log; exit 1; }
  grep -v amdgpu $OUT/sumregs_time.log
  kt kt_sumregs python3 tools/gpu_sumregs_time.py
  pmc pmc_sumregs_sq1 "$SQ1" python3 tools/sumregs_pmc_probe.py 400
  pmc pmc_sumregs_sq2 "$SQ2" python3 tools/sumregs_pmc_probe.py 400
  ;;
eval128)
  python3 tools/eval_once.py > $OUT/eval_128.log 2>&1 || { tail -20 $OUT/eval_128.log; exit 1; }
  grep -v amdgpu $OUT/eval_128.log
  kt kt_eval128 python3 tools/eval_once.py
  pmc pmc_eval128_sq1 "$SQ1" python3 tools/eval_once.py
  pmc pmc_eval128_sq2 "$SQ2" python3 tools/eval_once.py
  ;;
evalcfg5)
  python3 tools/eval_cfg5.py 8 3 400 > $OUT/eval_cfg5.log 2>&1 || { tail -20 $OUT/eval_cfg5.log; exit 1; }
  python3 tools/eval_cfg5.py 1 2 400 >> $OUT/eval_cfg5.log 2>&1 || { tail -20 $OUT/eval_cfg5.log; exit 1; }
  grep -v amdgpu $OUT/eval_cfg5.log
  kt kt_evalcfg5 python3 tools/eval_cfg5.py 8 1 400
  pmc pmc_evalcfg5_sq1 "$SQ1" python3 tools/eval_cfg5.py 8 1 40
  pmc pmc_evalcfg5_sq2 "$SQ2" python3 tools/eval_cfg5.py 8 1 40
  ;;
nd)
  tools/_bin/nd_unit time 1024 8 > $OUT/nd_unit_time.log 2>&1 || { tail $OUT/nd_unit_time.log; exit 1; }
  tools/_bin/nd_unit time 1024 1 >> $OUT/nd_unit_time.log 2>&1
  tools/_bin/nd_unit time 128 10 >> $OUT/nd_unit_time.log 2>&1
  tools/_bin/nd_unit time 128 10 sr >> $OUT/nd_unit_time.log 2>&1
  cat $OUT/nd_unit_time.log
  kt kt_nd tools/_bin/nd_unit time 1024 8
  ;;
esac
done
python3 tools/refresh_profiles.py aggregate $OUT
du -sh $OUT
