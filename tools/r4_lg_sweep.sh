#!/bin/bash
# the lane group of a read (KVQ_LG: 1 two lanes, 2 four, 3 eight, -1 per tile) on the headline input
# usage (through gpurun, repo root): bash tools/r4_lg_sweep.sh
cd ${GRAFT_REPO_ROOT:-.}
for lg in 2 1 3 -1 2 1; do
  export KVQ_LG=$lg
  python3 bench.py --no-cpu-baseline --no-end-to-end --steps 20 > /tmp/lg.json 2>/tmp/lg.err
  python3 -c "
import json,sys;d=json.load(open('/tmp/lg.json'));print('KVQ_LG', sys.argv[1], 'kernel %.4f ms  step %.4f ms  hits %d' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['config']['hits_per_step']))" $lg
done
