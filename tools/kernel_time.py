#!/usr/bin/env python3
"""Average time of the scan kernel on the bench workload, without any check of the results
(for the diagnostic KVQ_DBG switches, which break them on purpose).

usage: [KVQ_DBG=..] python tools/kernel_time.py [reads] [steps]
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
dg = scan.DeviceBuffer(g.nbytes); dg.upload(g); dd = scan.DeviceBuffer(n * rb)
_lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes)
co = b.analytic_chunk_offsets(n, rb, L)
t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
s = scan.Scanner(t)
ms = []
for rep in range(steps + 30):
    s.reset(); s.scan_device(dd.ptr, n * rb, co); r = s.finish(hits=False)
    if rep >= 30: ms.append(r['main_kernel_ms'])
print('KVQ_DBG=%s  kernel %.4f ms (min %.4f)  hits %d' % (os.environ.get('KVQ_DBG', '0'), float(np.mean(ms)), float(np.min(ms)), r['n_hits'] if 'n_hits' in r else -1))
