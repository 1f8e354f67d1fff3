#!/bin/bash
# the long reads' matcher grid (KVQ_MGRID) against the step of the realistic input with one 6 kB read in 100 000 records
cd ${GRAFT_REPO_ROOT:-.}
echo "no long reads:"; timeout -k 10 200 python3 tools/realistic_bench.py 3000000 2>&1 | grep "main kernel\|three steps"
for g in 0 64 128 256 512 1024; do
  if [ $g = 0 ]; then unset KVQ_MGRID; else export KVQ_MGRID=$g; fi
  echo "KVQ_MGRID=$g:"; timeout -k 10 200 python3 tools/realistic_bench.py 3000000 100000 2>&1 | grep "main kernel\|three steps"
done
