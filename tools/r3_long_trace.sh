#!/bin/bash
# kernel trace of tools/realistic_bench.py with one 6 kB read in LONG_EVERY records (0: none): what the redo of skipped tiles costs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_long; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/realistic_bench.py 3000000 ${LONG_EVERY:-100000} > $O/t.log 2>&1
f=$(find $O/t -name "*kernel_trace.csv" | head -1); python3 - $f <<'PY'
import csv,sys,collections,statistics
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): d[r['Kernel_Name'].split('(')[0][:44]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('kernel                                        launches  median us   (the first launches of a scan object that has not seen a skipped tile yet use small grids)')
for k,v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1])*len(kv[1]))[:14]: print('%-46s %6d %10.1f' % (k, len(v), statistics.median(v)))
PY
grep "path\|main kernel\|three steps" $O/t.log
