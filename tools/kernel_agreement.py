#!/usr/bin/env python3
"""rocprofv3's durations of the scan kernel against bench.py's own HIP-event figure of the same run.
usage: python tools/kernel_agreement.py <rocprof dir of a bench run> <that run's stdout log> [timed steps, default 5]"""
import csv, glob, json, sys
d, log = sys.argv[1], sys.argv[2]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = sorted((r for r in csv.DictReader(open(f)) if 'kvq_scan_bp' in r['Kernel_Name']), key=lambda r: int(r['Start_Timestamp']))
du = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
line = [l for l in open(log) if l.startswith('{"metric"')][-1]
b = json.loads(line)
print('%d launches of kvq_scan_bp in the run (preheat + warm-up + %d timed steps): rocprofv3 average of all %.4f ms (what the --stats csv shows; the first ten, at cold clocks, %.4f ms)' % (
    len(du), k, sum(du) / len(du), sum(du[:10]) / 10))
print('the %d timed launches: rocprofv3 %.4f ms (min %.4f, max %.4f); bench.py, HIP events around the same launches: %.4f ms  -> %.1f %% apart' % (
    k, sum(du[-k:]) / k, min(du[-k:]), max(du[-k:]), b['roofline']['avg_launch_ms'], abs(sum(du[-k:]) / k / b['roofline']['avg_launch_ms'] - 1) * 100))
