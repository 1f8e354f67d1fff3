#!/bin/bash
# bytes a tile owns (KVQ_TILE) against kernel and step time of the 1 M-read input (configs[1]), one and three steps in flight
# usage (through gpurun, repo root): bash tools/r4_tile_sweep.sh
cd ${GRAFT_REPO_ROOT:-.}
for t in 0 36000 32000 28000 24000 20000 16000; do
  for p in 1 3; do
    if [ $t = 0 ]; then unset KVQ_TILE; else export KVQ_TILE=$t; fi
    python3 bench.py --no-cpu-baseline --reads 1000000 --steps 50 --pipeline $p > /tmp/ts.json 2>/tmp/ts.err
    python3 -c "
import json,sys;d=json.load(open('/tmp/ts.json'));print('tile', sys.argv[1], 'pipeline', sys.argv[2], 'kernel %.4f ms  step %.4f ms  frac %.4f hits %d' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['hits_per_step']))" $t $p
  done
done
