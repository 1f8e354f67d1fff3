#!/bin/bash
# Round 4's fuzz campaign (tools/fuzz_parity.py: generated FastQ files, tables cut from them, random settings; GPU engine == oracle bit
# for bit) on the final sources: the default kernels, and the short-seed / draining / in-place-verification paths forced.
# usage (through gpurun, repo root): bash tools/r4_fuzz.sh <out file> [cases per run, default 1500]
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${1:-$R/gpurun_out/fuzz.txt}; N=${2:-1500}
cd $R
: > $OUT
run() { echo "== $1 seeds from $2 ($3)" >> $OUT; env $1 timeout -k 10 600 python3 tools/fuzz_parity.py $2 $N $3 2>&1 | tail -3 >> $OUT; }
run "KVQ_NONE=1" 2000000 seeded
run "KVQ_NONE=1" 2100000 general
run "KVQ_K=5" 2200000 seeded
run "KVQ_K=6" 2300000 seeded
run "KVQ_K=7" 2400000 seeded
run "KVQ_DENSE=1" 2500000 seeded
run "KVQ_SURVIVORS=0" 2600000 seeded
run "KVQ_DENSE=1 KVQ_SURVIVORS=0" 2700000 general
cat $OUT
