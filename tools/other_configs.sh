#!/bin/bash
# The configurations next to the headline one (BASELINE.json configs[1], [3], [4] and the dense-table sweep), one step at a
# time and with three in flight.  usage (through gpurun, repo root): bash tools/other_configs.sh [tag under gpurun_out/]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-round3}; mkdir -p $O; cd $R
: > $O/other_configs.txt
for p in 1 3; do
echo "== --pipeline $p (1: one step at a time, as in round 1's table; 3: the bench default)" >> $O/other_configs.txt
for args in "--reads 1000000" "--reads 40000000" "--readlen 300 --table MTBC+barcodes" "--readlen 100" "--readlen 250 --reads 6000000" "--table-scale 8 --reads 4000000" "--table-scale 32 --reads 1000000" "--table-scale 32 --reads 4000000"; do
  timeout -k 10 300 python3 bench.py $args --pipeline $p --no-cpu-baseline --no-end-to-end --steps 5 > $O/cfg.json 2> $O/cfg.err || { echo "$args: bench.py failed (see cfg.err)" >> $O/other_configs.txt; continue; }
  python3 - "$args" $O/cfg.json >> $O/other_configs.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); r = d['roofline']
print('%-44s step %.3f ms  kernel %.3f ms/launch (%d launch/step, %.3f GB)  roofline %.4f  %.3f G reads/s' % (
    sys.argv[1], d['ms_per_step'], r['avg_launch_ms'], r['launches_per_step'], r['algorithmic_bytes_per_launch'] / 1e9, r['frac'], d['value'] / 1e9))
PY
done
done
cat $O/other_configs.txt
