#!/usr/bin/env python3
"""Host-side view of one bench step: wall time of reset / scan_device / finish (device-resident 10 M reads).

usage: python tools/step_host_times.py [reads]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kvarq_amd import _lib, scan, synth
import bench

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    L = 150; rb = synth.record_bytes(L)
    L_ = _lib.lib()
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g, 'MTBC'))
    d_genome = scan.DeviceBuffer(g.nbytes); d_genome.upload(g)
    d_data = scan.DeviceBuffer(n * rb)
    assert L_.kvq_synth_reads_device(d_data.ptr, 0, n, L, synth.SEED, d_genome.ptr, g.nbytes) == 0
    offs = bench.analytic_chunk_offsets(n, rb, L)
    table = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
    sc = scan.Scanner(table)
    acc = np.zeros(4)
    for it in range(12):
        t0 = time.perf_counter(); sc.reset()
        t1 = time.perf_counter(); sc.scan_device(d_data.ptr, n * rb, offs, fpos_base=0)
        t2 = time.perf_counter(); r = sc.finish(hits=False, stats=False)
        t3 = time.perf_counter()
        if it >= 2: acc += [t1 - t0, t2 - t1, t3 - t2, r['main_kernel_ms'] * 1e-3]
    acc /= 10
    print('reset %.1f us  scan_device (enqueue) %.1f us  finish %.1f us  | main kernel %.1f us  | step %.1f us' %
          (acc[0] * 1e6, acc[1] * 1e6, acc[2] * 1e6, acc[3] * 1e6, acc[:3].sum() * 1e6))

if __name__ == '__main__':
    main()
