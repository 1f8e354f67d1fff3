#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_small; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 250000 1000000; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$n -- python3 $R/bench.py --preheat 0 --steps 10 --warmup 3 --no-cpu-baseline --pipeline 1 --reads $n > $O/t$n.log 2>&1
  echo "== reads $n"; f=$(find $O/t$n -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print('%-60s calls %5s avg %10.1f us  total %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
  tail -1 $O/t$n.log | cut -c1-300
done
