#!/bin/bash
# first run of kvq_scan_pool: parity tests, then kernel time against kvq_scan_bp (KVQ_KERNEL=v2)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_first
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for round in 1 2; do
  for k in v2 pool; do
    KVQ_KERNEL=$k timeout -k 10 200 python3 tools/kernel_time.py 10000000 20 2>&1 | tail -1 | sed "s/^/$k /"
  done
done | tee $O/kernel_time.txt
