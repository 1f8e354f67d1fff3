#!/usr/bin/env python3
"""One scan against the MTBC table scaled k-fold: which path it takes and how long the kernels run.
usage: python tools/r3_dense.py [scale] [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kvarq_amd import _lib, scan, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
L = 150; rb = synth.record_bytes(L)
g = synth.genome(); seqs = synth.both_strands(synth.table(g, 'MTBC', scale=k))
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g); dd = scan.DeviceBuffer(n * rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
import importlib.util
spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
co = b.analytic_chunk_offsets(n, rb, L)
t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
s = scan.Scanner(t)
for rep in range(3):
    s.reset(); t0 = time.perf_counter(); s.scan_device(dd.ptr, n * rb, co); r = s.finish(hits=False); dt = time.perf_counter() - t0
    print('x%d: %d sequences, %d reads: path %s  main kernels %.3f ms  step %.3f ms  hits %d' % (k, len(seqs), n, r['path'], r['main_kernel_ms'], dt * 1e3, r['n_hits']))
