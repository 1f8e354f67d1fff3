#!/usr/bin/env python3
"""The last steps of a rocprofv3 kernel trace, kernel by kernel with its hardware queue: what runs beside which scan
(usage: python tools/r4_trace_tail.py <..._kernel_trace.csv> [scans from the end, default 5])."""
import csv,sys
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:28],r.get('Queue_Id','?')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
scans=[i for i,r in enumerate(rows) if 'kvq_scan_bp' in r[2]]
back=int(sys.argv[2]) if len(sys.argv)>2 else 5       # start at the back-th scan kernel from the end, three scans' worth
i0=scans[-back]; t0=rows[i0][0]
for r in rows[i0:scans[-back+2]+1]:
    print('%9.1f %8.1f  q%-3s %s' % ((r[0]-t0)/1e3,(r[1]-r[0])/1e3,r[3],r[2]))
