#!/bin/bash
# HBM-side read bytes of the scan kernel with phases switched off (KVQ_DBG 0: whole kernel, 2: no filter and verify,
# 32: front end only -- no trim, so none of the per-record '@' / '+' byte loads); 5 M reads per launch.
# usage (through gpurun, repo root): bash tools/fetch_by_phase.sh <tag>
set -u
TAG=${1:-fetchph}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --preheat 0 --steps 2 --warmup 1 --no-cpu-baseline --reads 5000000"
: > $O/fetch_by_phase.txt
for d in ${DBGS:-0 2 32}; do
  KVQ_DBG=$d timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$d -- $B > $O/f$d.log 2>&1
  echo "== KVQ_DBG=$d (FETCH_SIZE in KB; x 2048 = bytes of wide streaming reads; algorithmic: 1.625e9 bytes)" >> $O/fetch_by_phase.txt
  python3 $R/tools/pmc_sum.py $O/f$d kvq_scan >> $O/fetch_by_phase.txt
  rm -rf $O/f$d
done
cat $O/fetch_by_phase.txt
