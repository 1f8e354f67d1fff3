#!/bin/bash
# scan-kernel grid (workgroup slots left free for the small kernels of the step in front) against kernel and step time, three steps in flight
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for g in 0 1024 1008 992 976 960 944; do
  if [ $g = 0 ]; then unset KVQ_GRID; else export KVQ_GRID=$g; fi
  python3 bench.py --no-cpu-baseline --steps 30 > /tmp/gs.json 2>/tmp/gs.err
  python3 -c "
import json,sys;d=json.load(open('/tmp/gs.json'));print('grid', sys.argv[1], 'kernel %.4f ms  step %.4f ms  all-kernels %.4f  hits %d' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline'].get('all_kernels_ms_per_step') or 0, d['config']['hits_per_step']))" $g
done
done
