#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for n in 250000 500000 1000000 2000000 4000000 10000000; do
  timeout -k 10 200 python3 tools/kernel_time.py $n 20 2>&1 | tail -1 | sed "s/^/reads $n /"
done
for gridv in 256 512 768; do KVQ_GRID=$gridv timeout -k 10 200 python3 tools/kernel_time.py 1000000 20 2>&1 | tail -1 | sed "s/^/reads 1000000 grid $gridv /"; done
for t in 20000 30000; do KVQ_TILE=$t timeout -k 10 200 python3 tools/kernel_time.py 1000000 20 2>&1 | tail -1 | sed "s/^/reads 1000000 tile $t /"; done
