#!/usr/bin/env python3
"""Round 4: what do the TRUE hits cost?  The bench workload scanned twice: reads sampled from the genome the table was cut from
(31 627 hits per 10 M reads) and reads sampled from another random genome (same text statistics, same false candidates per
read, no hit) -- the difference is the price of 0.003 hits per read (the byte-exact compare + emit of verify_item_bp).

usage: python tools/r4_nohits.py [reads] [steps]
"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from kvarq_amd import _lib, scan, synth
import importlib.util
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = 150; rb = synth.record_bytes(L)
g = synth.genome(); seqs = synth.both_strands(synth.table(g))
t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
co = b.analytic_chunk_offsets(n, rb, L)
dd = scan.DeviceBuffer(n * rb)
for name, gg in (('reads from the table\'s genome', g), ('reads from another genome', synth.genome(seed=synth.SEED + 12345))):
    dg = scan.DeviceBuffer(gg.nbytes); dg.upload(gg)
    _lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, gg.nbytes)
    s = scan.Scanner(t)
    ms = []
    for rep in range(steps + 30):
        s.reset(); s.scan_device(dd.ptr, n * rb, co); r = s.finish(hits=False)
        if rep >= 30: ms.append(r['main_kernel_ms'])
    print('%-32s kernel %.4f ms (min %.4f)  hits %d' % (name, float(np.mean(ms)), float(np.min(ms)), r['n_hits']))
    del s
