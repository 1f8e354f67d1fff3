#!/bin/bash
# kernel time of every library build under kvarq_amd/ab/ (two rounds, alternating), by lanes per read.
# usage (through gpurun, repo root): bash tools/r3_ab.sh "<KVQ_LG values>" [reads]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
LGS=${1:-"1 2"}; N=${2:-10000000}
cp kvarq_amd/libkvarq_hip.so /tmp/lib_orig.so
for round in 1 2; do
  for f in kvarq_amd/ab/*.so; do
    cp $f kvarq_amd/libkvarq_hip.so
    for lg in $LGS; do
      KVQ_LG=$lg timeout -k 10 200 python3 tools/kernel_time.py $N 20 2>&1 | tail -1 | sed "s/^/$(basename $f) LG=$lg /"
    done
  done
done
cp /tmp/lib_orig.so kvarq_amd/libkvarq_hip.so
KVQ_KERNEL=v2 timeout -k 10 200 python3 tools/kernel_time.py $N 20 2>&1 | tail -1 | sed "s/^/v2 /"
