#!/usr/bin/env python3
"""Timeline of the kernels of one bench step from a rocprofv3 --kernel-trace CSV.

usage: python tools/trace_step.py <dir with *_kernel_trace.csv>
Prints the kernels of the last complete step (between two kvq_scan_seeded bursts) with
start offsets, durations and the idle gap in front of each.
"""
import csv, glob, sys

def main():
    root = sys.argv[1]
    files = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # the last 4 main launches and everything between the first of them and the end of the step
    mains = [i for i, r in enumerate(rows) if ('kvq_scan_seeded' in r[2] or 'kvq_scan_bp' in r[2])]
    if len(mains) < 4:
        print('too few launches'); return
    lo = mains[-int(sys.argv[2]) if len(sys.argv) > 2 else -2]
    # walk back to the memsets/expand kernels that belong to the first batch
    while lo > 0 and rows[lo - 1][0] > rows[mains[(-int(sys.argv[2]) if len(sys.argv) > 2 else -2) - 1]][1] and not rows[lo - 1][2].startswith('kvq_gather'):
        lo -= 1
    t0 = rows[lo][0]; prev_end = t0
    for s, e, n in rows[lo:]:
        print('%9.1f us  +%7.1f gap  %8.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n[:70]))
        prev_end = max(prev_end, e)
    print('span %.1f us' % ((prev_end - t0) / 1e3))

if __name__ == '__main__':
    main()
