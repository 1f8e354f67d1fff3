#!/bin/bash
# Round 4: the SQ counters of every library build under kvarq_amd/ab/ on one box (one PMC pass each, 10 M reads per launch).
# usage (through gpurun, repo root): bash tools/r4_pmc_ab.sh <tag>
set -u
TAG=${1:-pmcab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cp $R/kvarq_amd/libkvarq_hip.so /tmp/lib_orig.so
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --preheat 0 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --pipeline 1"
: > $O/pmc_ab.txt
for f in $R/kvarq_amd/ab/*.so; do
  cp $f $R/kvarq_amd/libkvarq_hip.so
  n=$(basename $f .so)
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/$n -- $B > $O/$n.log 2>&1
  echo "== $n" >> $O/pmc_ab.txt
  python3 $R/tools/pmc_sum.py $O/$n kvq_scan_bp >> $O/pmc_ab.txt
  rm -rf $O/$n
done
cp /tmp/lib_orig.so $R/kvarq_amd/libkvarq_hip.so
cat $O/pmc_ab.txt
