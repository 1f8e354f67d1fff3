#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for k in v2 pool; do for st in 0 1 2 3; do
  KVQ_STAGGER=$st KVQ_LG=2 KVQ_KERNEL=$k timeout -k 10 200 python3 tools/kernel_time.py 10000000 15 2>&1 | tail -1 | sed "s/^/$k stagger $st /"
done; done
