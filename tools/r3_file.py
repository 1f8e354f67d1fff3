#!/usr/bin/env python3
"""End-to-end rate of engine.findseqs on a page-cache-warm plain FastQ file of the bench's shape
(host read + PCIe + kernels + the Python result).  usage: python tools/r3_file.py [reads] [nthreads ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kvarq_amd import _lib, engine, scan, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nts = [int(x) for x in sys.argv[2:]] or [4, 16]
L = 150; rb = synth.record_bytes(L)
g = synth.genome(); seqs = synth.both_strands(synth.table(g))
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g); dd = scan.DeviceBuffer(n * rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
path = '/tmp/kvq_e2e.fastq'
dd.download().tofile(path); dd.free()
for nt in nts:
    engine.config(maxerrors=2, minoverlap=25, minreadlength=25, Amin='.', nthreads=nt)
    for rep in range(3):
        t0 = time.perf_counter(); r = engine.findseqs(path, seqs); dt = time.perf_counter() - t0
        if os.environ.get('KVQ_TIMING'): sys.stderr.write('whole call %.1f ms\n' % (dt * 1e3))
    print('plain file %.2f GB  nthreads=%2d  %.3f s  %.1f M reads/s  %.2f GB/s  hits=%d' % (n * rb / 1e9, nt, dt, n / dt / 1e6, n * rb / dt / 1e9, len(r['hits'])))
os.remove(path)
