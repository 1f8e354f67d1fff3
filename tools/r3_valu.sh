#!/bin/bash
# SQ_INSTS_VALU / SALU / busy cycles of the scan kernel under a list of KVQ_DBG values.
# usage: bash tools/r3_valu.sh <tag> <label> "<dbg list>"   (the label only names the output; the library has one scan kernel)
set -u
TAG=$1; LABEL=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --preheat 0 --steps 2 --warmup 1 --no-cpu-baseline --reads 5000000"
: > $O/valu.txt
for d in $3; do
  KVQ_DBG=$d timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p$d -- $B > $O/p$d.log 2>&1
  echo "== $LABEL KVQ_DBG=$d" >> $O/valu.txt
  python3 $R/tools/pmc_sum.py $O/p$d kvq_scan >> $O/valu.txt
  rm -rf $O/p$d
done
cat $O/valu.txt
