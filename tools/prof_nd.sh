#!/bin/bash
# Kernel trace + two SQ counter passes of tools/_bin/nd_unit time M nimg (round 4: the wave-per-front factorisation).
# usage: tools/prof_nd.sh TAG M NIMG [sr] [leaf]   -> gpurun_out/TAG/{kt,sq1,sq2}/...  (summaries printed)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
SQ2="SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- tools/_bin/nd_unit time "$@" > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
python3 tools/refresh_profiles.py aggregate $OUT/kt
rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/sq1 -- tools/_bin/nd_unit time "$@" > $OUT/sq1.log 2>&1 || { tail -5 $OUT/sq1.log; exit 1; }
python3 tools/refresh_profiles.py aggregate $OUT/sq1
rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/sq2 -- tools/_bin/nd_unit time "$@" > $OUT/sq2.log 2>&1 || { tail -5 $OUT/sq2.log; exit 1; }
python3 tools/refresh_profiles.py aggregate $OUT/sq2
