#!/usr/bin/env python3
"""Where a short launch of the scan kernel spends its time: the instrumented build (KVQ_DBG=16) stamps every
workgroup's entry, end of prologue, end of first tile and exit on the constant 100 MHz clock.

usage: KVQ_DBG=16 python tools/r3_life.py [reads]
"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from kvarq_amd import _lib, scan, synth
import importlib.util
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
L = 150; rb = synth.record_bytes(L)
g = synth.genome(); seqs = synth.both_strands(synth.table(g))
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g); dd = scan.DeviceBuffer(n * rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
co = b.analytic_chunk_offsets(n, rb, L)
t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
s = scan.Scanner(t)
for rep in range(4):
    s.reset(); s.scan_device(dd.ptr, n * rb, co); r = s.finish(hits=False)
c = [int(x) for x in r['counters'][4 + 916:4 + 924].astype(np.uint64)]
wgs = c[3]; tick = 0.01   # us per tick of the 100 MHz clock
first_in = (~c[6]) & 0xFFFFFFFFFFFFFFFF
print('reads %d  main kernel %.1f us  workgroups %d' % (n, r['main_kernel_ms'] * 1e3, wgs))
print('per workgroup (average): prologue %.1f us, first tile %.1f us, entry to end of tile loop %.1f us, epilogue %.1f us' % (c[0] / wgs * tick, c[1] / wgs * tick, c[2] / wgs * tick, c[7] / wgs * tick))
print('first entry to last exit %.1f us; first entry to last entry %.1f us' % ((c[4] - first_in) * tick, (c[5] - first_in) * tick))
