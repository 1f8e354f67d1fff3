#!/bin/bash
# A/B of library builds: kernel time (two rounds) and instruction counts of the production kernel.
# usage: bash tools/r3_ab2.sh "<KVQ_LG values>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
LGS=${1:-"2"}
bash tools/r3_ab.sh "$LGS"
for f in kvarq_amd/ab/*.so; do
  cp $f kvarq_amd/libkvarq_hip.so
  for lg in $LGS; do
    KVQ_LG=$lg bash tools/r3_valu.sh r3_ab2 pool "0" > /dev/null 2>&1
    echo "== $(basename $f) LG=$lg: $(grep 'INSTS_VALU\|INSTS_SALU\|INSTS_LDS\|BUSY' gpurun_out/r3_ab2/valu.txt | awk '{printf "%s %.1fM  ", $1, $2/1e6}')"
  done
done
KVQ_KERNEL=v2 bash tools/r3_valu.sh r3_ab2 v2 "0" > /dev/null 2>&1
echo "== v2: $(grep 'INSTS_VALU\|INSTS_SALU\|INSTS_LDS\|BUSY' gpurun_out/r3_ab2/valu.txt | awk '{printf "%s %.1fM  ", $1, $2/1e6}')"
