#!/usr/bin/env python3
"""Eight scans enqueued back to back on eight scan objects, all finished afterwards: the idle time between consecutive scan
kernels when no `finish` (its kernels, its copies, its host wait) is in the way.  usage: python tools/r3_gap2.py [reads]"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from kvarq_amd import _lib, scan, synth
import importlib.util
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 150; rb = synth.record_bytes(L)
g = synth.genome(); seqs = synth.both_strands(synth.table(g))
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g); dd = scan.DeviceBuffer(n * rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
co = b.analytic_chunk_offsets(n, rb, L)
t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
ring = [scan.Scanner(t) for _ in range(8)]
Lb = _lib.lib()
for rep in range(4):
    for sc in ring: sc.reset()
    Lb.kvq_device_synchronize(); t0 = time.perf_counter()
    for sc in ring: sc.scan_device(dd.ptr, n * rb, co)
    t1 = time.perf_counter()
    rs = [sc.finish(hits=False, stats=False) for sc in ring]
    dt = time.perf_counter() - t0
    gaps = [Lb.kvq_scan_gap_ms(ring[i].h, ring[i + 1].h) * 1e3 for i in range(7)]
    print('8 scans enqueued in %.2f ms, all done after %.2f ms (%.3f ms each); scan kernel %.3f ms; idle between consecutive scan kernels (us): %s' % (
        (t1 - t0) * 1e3, dt * 1e3, dt / 8 * 1e3, float(np.median([r['main_kernel_ms'] for r in rs])), ' '.join('%.0f' % x for x in gaps)))
