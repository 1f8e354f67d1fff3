#!/bin/bash
# effective shader clock of the scan kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-clk}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/g -- python3 $R/bench.py --preheat 30 --steps 5 --warmup 1 --no-cpu-baseline > $O/g.log 2>&1
python3 - $O/g <<'PY'
import csv, glob, sys
cnt = {}; dur = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'kvq_scan' in r['Kernel_Name']: cnt.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'kvq_scan' in r['Kernel_Name']: dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
ks = sorted(k for k in cnt if k in dur)[-5:]
for k in ks: print('dispatch %s: %.4f ms  GRBM_GUI_ACTIVE %.0f -> %.3f GHz' % (k, dur[k] * 1e3, cnt[k].get('GRBM_GUI_ACTIVE', 0), cnt[k].get('GRBM_GUI_ACTIVE', 0) / 8 / dur[k] / 1e9))
PY
rm -rf $O/g
