#!/usr/bin/env python3
"""Kernel timeline (start, end, queue) between the last scan kernels of a rocprofv3 --kernel-trace csv.
usage: python tools/r3_timeline.py <kernel_trace.csv> [scans back, default 4] [scans shown, default 2]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 4
shown = int(sys.argv[3]) if len(sys.argv) > 3 else 2
idx = [i for i, r in enumerate(rows) if 'kvq_scan_bp' in r['Kernel_Name']]
i0, i1 = idx[-back], idx[-back + shown]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-24s q%s start %8.1f end %8.1f dur %7.1f' % (r['Kernel_Name'].split('(')[0][:24], r['Queue_Id'], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
