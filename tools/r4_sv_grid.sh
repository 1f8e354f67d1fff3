#!/bin/bash
# workgroups of the survivors' kernel beside a scan (KVQ_SV_GRID) against kernel and step time of the headline
# usage (through gpurun, repo root): bash tools/r4_sv_grid.sh
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for g in 64 32 16 128; do
  export KVQ_SV_GRID=$g
  python3 bench.py --no-cpu-baseline --no-end-to-end --steps 30 > /tmp/sv.json 2>/tmp/sv.err
  python3 -c "
import json,sys;d=json.load(open('/tmp/sv.json'));print('KVQ_SV_GRID', sys.argv[1], 'kernel %.4f ms  step %.4f ms  all-kernels %.4f  hits %d' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline'].get('all_kernels_ms_per_step') or 0, d['config']['hits_per_step']))" $g
done
done
