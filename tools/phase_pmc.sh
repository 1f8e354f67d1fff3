#!/bin/bash
# Per-phase instruction counts of the scan kernel: PMC passes with the phases switched off one by one
# (KVQ_DBG 0: whole kernel, 1: no verify, 2: no filter and verify, 32: front end only); 5 M reads per launch.
# usage (through gpurun, repo root): bash tools/phase_pmc.sh <tag> [kvarq_amd/ab/<name>.so]
set -u
TAG=${1:-ph}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ $# -ge 2 ]; then cp $R/$2 $R/kvarq_amd/libkvarq_hip.so; fi
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --preheat 0 --steps 2 --warmup 1 --no-cpu-baseline --reads 5000000"
: > $O/pmc_phases.txt
for d in 0 1 2 32; do
  KVQ_DBG=$d timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/ph$d -- $B > $O/ph$d.log 2>&1
  echo "== KVQ_DBG=$d" >> $O/pmc_phases.txt
  python3 $R/tools/pmc_sum.py $O/ph$d kvq_scan >> $O/pmc_phases.txt
  rm -rf $O/ph$d
done
cat $O/pmc_phases.txt
