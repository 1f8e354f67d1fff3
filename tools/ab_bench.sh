#!/bin/bash
# A/B timing of library builds on one GPU box: kvarq_amd/ab/<name>.so are copied over
# libkvarq_hip.so in turn (three rounds, alternating) and bench.py's kernel time is printed.
# usage (through gpurun, repo root): bash tools/ab_bench.sh [bench args]
cd ${GRAFT_REPO_ROOT:-.}
cp kvarq_amd/libkvarq_hip.so /tmp/lib_orig.so
for round in 1 2 3; do
  for f in kvarq_amd/ab/*.so; do
    cp $f kvarq_amd/libkvarq_hip.so
    python3 bench.py --no-cpu-baseline --steps 20 "$@" > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "
import json,sys;d=json.load(open('/tmp/ab.json'));print(sys.argv[1], 'kernel %.4f ms  step %.4f ms  all-kernels %.4f ms  hits %d' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline'].get('all_kernels_ms_per_step') or 0, d['config']['hits_per_step']))" $(basename $f)
  done
done
cp /tmp/lib_orig.so kvarq_amd/libkvarq_hip.so
