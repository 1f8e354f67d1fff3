#!/bin/bash
# step time of the default bench (three steps in flight) against the number of workgroups of the scan kernel: how many
# workgroup slots should a scan leave free for the other steps' small kernels?  (KVQ_GRID; 1024 = none free)
for g in ${1:-992 976 960 944 928}; do for i in 1 2 3; do
  KVQ_GRID=$g python3 bench.py --no-cpu-baseline --no-end-to-end --steps 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('KVQ_GRID=$g step %.4f ms kernel %.4f ms  %.3f G reads/s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value']/1e9))"
done; done
