#!/bin/bash
# Timing of one library build under several environments, alternating (three rounds).
# usage (through gpurun, repo root): bash tools/ab_env.sh <lib.so> "ENV1=.. ENV2=.." "ENV=.." ... [-- bench args]
cd ${GRAFT_REPO_ROOT:-.}
LIB=$1; shift
cp $LIB kvarq_amd/libkvarq_hip.so
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "${1:-}" == "--" ] && shift
for round in 1 2 3; do
  for e in "${ENVS[@]}"; do
    env $e python3 bench.py --no-cpu-baseline --steps 20 "$@" > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "
import json,sys;d=json.load(open('/tmp/ab.json'));print('%-40s kernel %.4f ms  step %.4f ms  hits %d' % (sys.argv[1], d['roofline']['avg_launch_ms'], d['ms_per_step'], d['config']['hits_per_step']))" "$e"
  done
done
