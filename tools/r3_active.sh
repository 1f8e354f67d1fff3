#!/bin/bash
# activity counters of the scan kernel: usage bash tools/r3_active.sh <tag> <KVQ_KERNEL> <KVQ_LG> [lib.so]
set -u
TAG=$1; export KVQ_KERNEL=$2; export KVQ_LG=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -ge 4 ] && cp $R/$4 $R/kvarq_amd/libkvarq_hip.so
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --preheat 0 --steps 2 --warmup 1 --no-cpu-baseline --reads 5000000"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/a -- $B > $O/a.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/b -- $B > $O/b.log 2>&1
echo "== $KVQ_KERNEL LG=$KVQ_LG ${4:-}"
python3 $R/tools/pmc_sum.py $O/a kvq_scan; python3 $R/tools/pmc_sum.py $O/b kvq_scan
rm -rf $O/a $O/b
