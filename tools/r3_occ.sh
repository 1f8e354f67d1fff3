#!/bin/bash
# time against occupancy, by phase: KVQ_GRID 1024 / 512 (8 / 4 waves per SIMD) x KVQ_DBG 32 / 130 / 2 / 1 / 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for d in 32 130 2 1 0; do for gr in 1024 512; do
  KVQ_GRID=$gr KVQ_DBG=$d timeout -k 10 200 python3 tools/kernel_time.py 10000000 15 2>&1 | tail -1 | sed "s/^/grid $gr /"
done; done
