#!/usr/bin/env python3
"""Per-phase cycles of the seed-filter kernel from the instrumented build (KVQ_DBG=16): in-kernel
s_memtime stamps of wave 0, summed per phase in spare counter slots.

usage: KVQ_DBG=16 python tools/phase_stamps.py
"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from kvarq_amd import _lib, scan, synth
g = synth.genome(); seqs = synth.both_strands(synth.table(g))
n, L = 2500000, 150; rb = synth.record_bytes(L)
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g); dd = scan.DeviceBuffer(n*rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
import importlib.util
spec = importlib.util.spec_from_file_location('bench', os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
co = b.analytic_chunk_offsets(n, rb, L)
t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
s = scan.Scanner(t)
for rep in range(2):
    s.reset(); s.scan_device(dd.ptr, n*rb, co); r = s.finish(hits=False)
c = r['counters'][4+900:4+908].astype(np.float64)
names = ['P0 store+bar', 'P1a scan+bar', 'P1b write+bar', 'P2', 'P3 desc+trim', 'P3 filter', 'P4a+P4b (own queue)', 'tile-end barrier']
tile_bytes = int(os.environ.get('KVQ_TILE', 39760))   # what kvq_choose_tile picks for 150 bp records
tot = c.sum()
wgs = int(os.environ.get('KVQ_WGS', 1024))
print('main kernel ms', r['main_kernel_ms'], 'tiles/WG', (n*rb/tile_bytes)/wgs)
for nm, v in zip(names, c): print('%-16s %6.1f%%  %8.0f cycles/tile' % (nm, 100*v/tot, v/(n*rb/tile_bytes)))
x = r['counters'][4+924:4+926].astype(np.float64) / (n*rb/tile_bytes)
print('of the first phase: wait at the tile-top barrier %.0f, wait for the text (vmcnt) %.0f cycles/tile; the rest is P0 and the newline scan' % (x[0], x[1]))
if os.environ.get('KVQ_KERNEL') != 'planes':
    w = r['counters'][4+908:4+916].astype(np.float64) / (n*rb/tile_bytes)
    print('P3+P4 cycles/tile by wave:', ' '.join('%.0f' % v for v in w))
t = r['counters'][4+930:4+938].astype(np.float64)
if t.sum() > 0:
    print('P4 tallies per read: anchor candidates %.3f, fixed-block candidates %.3f, index entries tested %.3f, passed the 16-base test %.5f' % (t[0]/n, t[1]/n, t[2]/n, t[3]/n))
    print('P4 per wave and stretch (%.0f stretches, %.2f reads each): candidate batches %.2f, rounds of the entry loop %.2f, byte-exact verifications %.3f' % (t[6], n/max(t[6],1), t[5]/max(t[6],1), t[4]/max(t[6],1), t[7]/max(t[6],1)))
