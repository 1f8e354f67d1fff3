#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp kvarq_amd/ab/p1s0.so kvarq_amd/libkvarq_hip.so
for lg in 1 2; do echo "== pool p1 LG=$lg"; KVQ_LG=$lg KVQ_DBG=16 timeout -k 10 200 python3 tools/phase_stamps.py 2>&1 | tail -10; done
cp kvarq_amd/ab/p0s0.so kvarq_amd/libkvarq_hip.so
for lg in 2; do echo "== pool p0 LG=$lg"; KVQ_LG=$lg KVQ_DBG=16 timeout -k 10 200 python3 tools/phase_stamps.py 2>&1 | tail -10; done
