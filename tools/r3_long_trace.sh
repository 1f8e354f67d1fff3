#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_long; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/realistic_bench.py 3000000 ${LONG_EVERY:-100000} > $O/t.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]: print('%-60s calls %5s avg %10.1f us  total %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
tail -4 $O/t.log
