#!/usr/bin/env python3
"""Idle time between consecutive scan kernels with three steps in flight, without a profiler attached: HIP events of
the library (kvq_scan_gap_ms: end of one step's scan kernel -> start of the next step's).  Four scan objects take
turns so that a step's events are still there when the step behind it has finished.

usage: python tools/r3_gap.py [reads] [steps]
"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from kvarq_amd import _lib, scan, synth
import importlib.util
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3                  # steps in flight
L = 150; rb = synth.record_bytes(L)
g = synth.genome(); seqs = synth.both_strands(synth.table(g))
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g); dd = scan.DeviceBuffer(n * rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
co = b.analytic_chunk_offsets(n, rb, L)
t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
ring = [scan.Scanner(t) for _ in range(depth + 1)]
Lb = _lib.lib()
gaps, kern, flying, last = [], [], [], None
t0 = None
for k in range(steps + 8):
    if k == 8: Lb.kvq_device_synchronize(); t0 = time.perf_counter(); gaps.clear(); kern.clear()
    sc = ring[k % (depth + 1)]; sc.reset(); sc.scan_device(dd.ptr, n * rb, co); flying.append(sc)
    if len(flying) == depth:
        f = flying.pop(0); r = f.finish(hits=False, stats=False); kern.append(r['main_kernel_ms'])
        if last is not None: gaps.append(Lb.kvq_scan_gap_ms(last.h, f.h) * 1e3)
        last = f
while flying:
    f = flying.pop(0); r = f.finish(hits=False, stats=False); kern.append(r['main_kernel_ms'])
    gaps.append(Lb.kvq_scan_gap_ms(last.h, f.h) * 1e3); last = f
dt = time.perf_counter() - t0
print('depth %d, %d steps: %.4f ms per step; scan kernel %.4f ms (median); idle between one step\'s scan kernel and the next one\'s: median %.1f us, mean %.1f, min %.1f, max %.1f' % (
    depth, steps, dt / steps * 1e3, float(np.median(kern)), float(np.median(gaps)), float(np.mean(gaps)), min(gaps), max(gaps)))
print('gaps (us):', ' '.join('%.0f' % x for x in gaps))
