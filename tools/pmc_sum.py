#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel name.

usage: python tools/pmc_sum.py <dir with *counter_collection.csv> [kernel substring]
Prints, per counter, the average value per launch of the kernels whose name contains the substring.
"""
import csv, glob, sys, collections

def main():
    root = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else 'kvq_scan_seeded'
    acc = collections.defaultdict(float); launches = collections.defaultdict(set)
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if sub not in r['Kernel_Name']: continue
                acc[r['Counter_Name']] += float(r['Counter_Value'])
                launches[r['Counter_Name']].add(r['Dispatch_Id'])
    for k in sorted(acc):
        n = max(1, len(launches[k]))
        print('%-28s %16.0f per launch (%d launches)' % (k, acc[k] / n, n))

if __name__ == '__main__':
    main()
