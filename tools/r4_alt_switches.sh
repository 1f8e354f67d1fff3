#!/bin/bash
# The GPU parity suite under every alternate code path the library has a switch for (each line: one whole run of `pytest -m gpu`).
# usage (through gpurun, repo root): bash tools/r4_alt_switches.sh <out file> [first switch index, default 0] [last index, default all]
# (22 runs of ~61 s: two gpurun calls; the second one with a start index)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${1:-$R/gpurun_out/alt_switches.txt}; FIRST=${2:-0}; LAST=${3:-99}
cd $R
SW=("KVQ_STRIDE=2" "KVQ_STRIDE=4" "KVQ_STRIDE=8" "KVQ_LG=-1" "KVQ_LG=1" "KVQ_LG=3" "KVQ_DBG=4" "KVQ_ORDER=mergesort" "KVQ_GRID=64" "KVQ_BLOCK_CACHE=0" "KVQ_KEEP_SCAN=0"
    "KVQ_K=5" "KVQ_K=6" "KVQ_K=7" "KVQ_DENSE=1" "KVQ_DENSE=0" "KVQ_SURVIVORS=0" "KVQ_DENSE=1 KVQ_SURVIVORS=0" "KVQ_SURV_CAP=40" "KVQ_K=6 KVQ_SURV_CAP=0" "KVQ_SV_MODE=1" "KVQ_SV_MODE=0")
: > $OUT
for ((i = FIRST; i < ${#SW[@]} && i <= LAST; i++)); do
  echo "== ${SW[$i]}" >> $OUT
  env ${SW[$i]} timeout -k 10 400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -1 >> $OUT
done
cat $OUT
