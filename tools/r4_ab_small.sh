#!/bin/bash
# A/B of library builds (kvarq_amd/ab/*.so) on the SMALL inputs -- the sequencer-like 3 M records with and without long reads, 1 M reads --
# and then on the headline (tools/ab_bench.sh)
# usage (through gpurun, repo root): bash tools/r4_ab_small.sh
cd ${GRAFT_REPO_ROOT:-.}
cp kvarq_amd/libkvarq_hip.so /tmp/lib_orig.so
for round in 1 2; do
  for f in kvarq_amd/ab/*.so; do
    cp $f kvarq_amd/libkvarq_hip.so
    echo "== $(basename $f)"
    python3 tools/realistic_bench.py 3000000 2>&1 | grep "main kernel\|three steps"
    python3 tools/realistic_bench.py 3000000 100000 2>&1 | grep "main kernel\|three steps"
    python3 bench.py --no-cpu-baseline --no-end-to-end --reads 1000000 --steps 50 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('1M reads: step %.4f kernel %.4f' % (d['ms_per_step'], r['avg_launch_ms']))"
  done
done
cp /tmp/lib_orig.so kvarq_amd/libkvarq_hip.so
bash tools/ab_bench.sh
