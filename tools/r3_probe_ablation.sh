#!/bin/bash
# Round 3: what the per-record '@' / '+' probes cost (KVQ_DBG=8 switches them off; results then miss the format errors).
# usage (through gpurun, repo root): bash tools/r3_probe_ablation.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_probe
mkdir -p $O
cd $R
for round in 1 2 3; do
  for d in 0 8; do
    KVQ_DBG=$d timeout -k 10 200 python3 tools/kernel_time.py 10000000 20 2>&1 | tail -1
  done
done | tee $O/kernel_time.txt
DBGS="0 8" bash tools/fetch_by_phase.sh r3_probe_fetch
