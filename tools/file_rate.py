#!/usr/bin/env python3
"""End-to-end rate of engine.findseqs on files (host read + PCIe + kernels), the
PCIe-inclusive number DESIGN.md quotes next to the HBM-resident bench value."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gzip  # noqa: E402

import numpy as np  # noqa: E402

from kvarq_amd import _lib, engine, scan, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
L = 150
rb = synth.record_bytes(L)
g = synth.genome()
seqs = synth.both_strands(synth.table(g))
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g)
dd = scan.DeviceBuffer(n * rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
data = dd.download()
path = '/tmp/kvq_rate.fastq'
data.tofile(path)
gzpath = '/tmp/kvq_rate_small.fastq.gz'
with open(gzpath, 'wb') as f:
    f.write(gzip.compress(data[:200_000 * rb].tobytes(), 1))
for nt in (1, 4, 16):
    engine.config(maxerrors=2, minoverlap=25, minreadlength=25, Amin='.', nthreads=nt)
    for rep in range(2):
        t0 = time.perf_counter()
        r = engine.findseqs(path, seqs)
        dt = time.perf_counter() - t0
    print('plain  nthreads=%2d  %.3f s  %.1f M reads/s  %.2f GB/s  hits=%d' % (nt, dt, n / dt / 1e6, n * rb / dt / 1e9, len(r['hits'])))
t0 = time.perf_counter()
r = engine.findseqs(gzpath, seqs)
dt = time.perf_counter() - t0
print('gzip   200k reads     %.3f s  %.2f M reads/s (single-threaded inflate)  hits=%d' % (dt, 0.2 / dt, len(r['hits'])))
# the same text bgzip'ed (BGZF blocks of 60000 bytes): block-parallel inflate
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_host_logic import bgzf  # noqa: E402
nb = 1_000_000
bpath = '/tmp/kvq_rate_bgzf.fastq.gz'
with open(bpath, 'wb') as f:
    f.write(bgzf(data[:nb * rb].tobytes(), level=1))
for nt in (1, 4, 16):
    engine.config(maxerrors=2, minoverlap=25, minreadlength=25, Amin='.', nthreads=nt)
    for rep in range(2):
        t0 = time.perf_counter()
        r = engine.findseqs(bpath, seqs)
        dt = time.perf_counter() - t0
    print('bgzf   nthreads=%2d  %.3f s  %.1f M reads/s  %.2f GB/s inflated  hits=%d' % (nt, dt, nb / dt / 1e6, nb * rb / dt / 1e9, len(r['hits'])))
# a pair of ordinary .gz files (reads_1 / reads_2): the second file's reader runs ahead of the stream
npair = 600_000
pp = ['/tmp/kvq_rate_1.fastq.gz', '/tmp/kvq_rate_2.fastq.gz']
for k, q in enumerate(pp):
    with open(q, 'wb') as f:
        f.write(gzip.compress(data[k * npair * rb:(k + 1) * npair * rb].tobytes(), 1))
outs = []
for nt, what in ((1, 'one reader, file after file'), (4, 'a reader per file')):
    engine.config(maxerrors=2, minoverlap=25, minreadlength=25, Amin='.', nthreads=nt)
    for rep in range(2):
        t0 = time.perf_counter()
        r = engine.findseqs(pp, seqs)
        dt = time.perf_counter() - t0
    outs.append((r['hits'], r['hitseqs'], r['stats']))
    print('gz pair nthreads=%2d  %.3f s  %.2f M reads/s  %.2f GB/s inflated (%s)  hits=%d' % (nt, dt, 2 * npair / dt / 1e6, 2 * npair * rb / dt / 1e9, what, len(r['hits'])))
assert outs[0] == outs[1]
os.remove(path); os.remove(gzpath); os.remove(bpath); os.remove(pp[0]); os.remove(pp[1])
