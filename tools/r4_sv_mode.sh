#!/bin/bash
# where the survivors' kernel runs: KVQ_SV_MODE unset = the library's choice (beside the next scan when jobs are in flight), 0 = always behind
# its own scan with the next one waiting, 1 = always beside the next scan
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for m in -1 0 1; do
  if [ $m = -1 ]; then unset KVQ_SV_MODE; else export KVQ_SV_MODE=$m; fi
  python3 bench.py --no-cpu-baseline --no-end-to-end --steps 30 > /tmp/sv.json 2>/tmp/sv.err
  python3 -c "
import json,sys;d=json.load(open('/tmp/sv.json'));print('KVQ_SV_MODE', sys.argv[1], 'kernel %.4f ms  step %.4f ms  all-kernels %.4f  hits %d' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline'].get('all_kernels_ms_per_step') or 0, d['config']['hits_per_step']))" $m
done
done
