#!/bin/bash
# Quick look at one library build on an MI355X box: phase stamps + two SQ counter passes.
# usage (through gpurun, repo root): bash tools/quick_prof.sh <tag> [kvarq_amd/ab/<name>.so]
set -u
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ $# -ge 2 ]; then cp $R/$2 $R/kvarq_amd/libkvarq_hip.so; fi
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --preheat 0 --steps 2 --warmup 1 --no-cpu-baseline"
KVQ_DBG=16 timeout -k 10 200 python3 $R/tools/phase_stamps.py > $O/phase_stamps.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
cd $R
python3 tools/pmc_sum.py $O/sq1 kvq_scan > $O/pmc.txt; python3 tools/pmc_sum.py $O/sq2 kvq_scan >> $O/pmc.txt
rm -rf $O/sq1 $O/sq2
cat $O/phase_stamps.txt $O/pmc.txt
