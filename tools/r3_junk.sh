#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for round in 1 2; do for f in kvarq_amd/ab/*.so; do cp $f kvarq_amd/libkvarq_hip.so
  for k in v2; do KVQ_LG=2 KVQ_KERNEL=$k timeout -k 10 200 python3 tools/kernel_time.py 10000000 15 2>&1 | tail -1 | sed "s/^/$(basename $f) $k /"; done
done; done
