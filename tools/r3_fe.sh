#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for d in 32 96 0 64; do
  KVQ_KERNEL=v2 KVQ_DBG=$d timeout -k 10 200 python3 tools/kernel_time.py 10000000 20 2>&1 | tail -1
done
for gridv in 512 768; do KVQ_GRID=$gridv KVQ_KERNEL=v2 KVQ_DBG=32 timeout -k 10 200 python3 tools/kernel_time.py 10000000 20 2>&1 | tail -1 | sed "s/^/grid $gridv /"; done
