#!/bin/bash
# kernel time + VALU count of kvq_scan_pool by lanes per read.  usage: bash tools/r3_lg.sh <tag>
set -u
TAG=${1:-r3_lg}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for lg in 1 2 0; do
  KVQ_LG=$lg timeout -k 10 200 python3 tools/kernel_time.py 10000000 20 2>&1 | tail -1 | sed "s/^/LG=$lg /"
done
for lg in 1 2; do
  KVQ_LG=$lg bash tools/r3_valu.sh ${TAG}_$lg pool "0 1 2 130 34" > /dev/null 2>&1
  echo "=== LG=$lg"; grep "==\|INSTS_VALU\|BUSY" gpurun_out/${TAG}_$lg/valu.txt
done
