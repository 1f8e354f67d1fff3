#!/bin/bash
# how much of the kernel's time is the way to HBM: KVQ_DBG=64 makes every tile read one of the first 64 tiles' text (L2 hits)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for k in v2 pool; do for d in 0 64; do
  KVQ_KERNEL=$k KVQ_DBG=$d timeout -k 10 200 python3 tools/kernel_time.py 10000000 20 2>&1 | tail -1 | sed "s/^/$k /"
done; done
KVQ_KERNEL=pool KVQ_LG=2 KVQ_DBG=64 timeout -k 10 200 python3 tools/kernel_time.py 10000000 20 2>&1 | tail -1 | sed "s/^/pool LG2 /"
