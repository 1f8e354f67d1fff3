#!/bin/bash
# per-phase counters + stamps of one kernel choice.  usage: bash tools/r3_prof.sh <tag> [KVQ_KERNEL value]
set -u
TAG=${1:-r3prof}; export KVQ_KERNEL=${2:-pool}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/phase_pmc.sh $TAG > /dev/null 2>&1
cat gpurun_out/$TAG/pmc_phases.txt
KVQ_DBG=16 timeout -k 10 200 python3 tools/phase_stamps.py 2>&1 | tee gpurun_out/$TAG/phase_stamps.txt
