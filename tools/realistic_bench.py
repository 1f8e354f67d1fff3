#!/usr/bin/env python3
"""Throughput on FastQ that looks like a sequencer's output rather than like the bench's fixed-size
records: Illumina-style headers of varying width, reads of 35..151 bases (adapter-trimmed mix),
qualities that decay towards the 3' end (many bytes below Amin), now and then an N.  Reports the
seed-filter kernel's rate on the device-resident text and whether any batch fell back to the
exhaustive kernels (it must not).

usage: python tools/realistic_bench.py [records, default 2000000] [long-every]

long-every N: one record in N is a 6 kB read (longer than any tile's look-ahead): its tile hands all its
records to the exhaustive kernels (path.tiles_rescanned) -- the step time says what that costs.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kvarq_amd import scan, synth


def make_text(n, g, seed=11, long_every=0):
    rng = np.random.default_rng(seed)
    out = []
    comp = np.zeros(256, dtype=np.uint8); comp[list(b'ACGT')] = list(b'TGCA')
    pos = rng.integers(0, len(g) - 160, size=n)
    full = rng.random(n) < 0.7
    lens = np.where(full, 151, rng.integers(35, 152, size=n))
    strand = rng.random(n) < 0.5
    for i in range(n):
        L = int(lens[i])
        if long_every and i % long_every == long_every // 2:
            L = 6000
            pos[i] = min(int(pos[i]), len(g) - L - 1)
        b = g[pos[i]:pos[i] + L]
        if strand[i]:
            b = comp[b[::-1]]
        b = b.copy()
        if i % 50 == 0:
            b[rng.integers(0, L)] = ord('N')
        # quality: Q36 plateau, decaying tail, a few dips
        q = np.full(L, 36 + 33, dtype=np.uint8)
        tail = int(rng.integers(0, L // 2))
        if tail:
            q[L - tail:] = np.clip(36 - (np.arange(tail) * rng.integers(10, 40) // max(1, tail)) - rng.integers(0, 8, size=tail), 2, 36) + 33
        dips = rng.integers(0, L, size=rng.integers(0, 4))
        q[dips] = 2 + 33
        hdr = b'@M0%d:%d:000000000-A%dK:1:%d:%d:%d 1:N:0:%d' % (1000 + i % 7, 40 + i % 13, i % 97, 1101 + i % 19, int(rng.integers(1000, 30000)), int(rng.integers(1000, 30000)), i % 96)
        out.append(hdr + b'\n' + b.tobytes() + b'\n+\n' + q.tobytes() + b'\n')
    return b''.join(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    long_every = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    base_n = min(n, 100_000)
    t0 = time.time()
    base = make_text(base_n, g, long_every=long_every)
    reps = max(1, n // base_n)
    text = np.frombuffer(base * reps, dtype=np.uint8)
    print('generated %d records, %.1f MB (%.1f s)' % (base_n * reps, text.nbytes / 1e6, time.time() - t0))
    co = scan.chunk_offsets(text)
    d = scan.DeviceBuffer(text.nbytes)
    d.upload(text)
    t = scan.Table(seqs, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
    s = scan.Scanner(t)
    for it in range(6):
        s.reset()
        t1 = time.perf_counter()
        s.scan_device(d.ptr, text.nbytes, co)
        r = s.finish(hits=False, stats=(it == 5))
        if it < 5: dt = time.perf_counter() - t1                  # (the timed step: without the statistics as Python objects, which the last round fetches)
    # the same with three jobs in flight, as bench.py runs them (the next scan is enqueued before the last one's results
    # are waited for: a tile's records that go round again through the host then cost GPU time, not idle time)
    ring = [s, scan.Scanner(t), scan.Scanner(t)]
    def run(k):
        flying = []
        for i in range(k):
            sc = ring[i % 3]; sc.reset(); sc.scan_device(d.ptr, text.nbytes, co); sc.finish_begin(); flying.append(sc)
            if len(flying) == 3:
                flying.pop(0).finish(hits=False, stats=False)
        while flying:
            flying.pop(0).finish(hits=False, stats=False)
    run(6)
    t1 = time.perf_counter(); run(12); dt3 = (time.perf_counter() - t1) / 12
    ms = r['main_kernel_ms']
    print('path', r['path'], ' hits', r['n_hits'], ' records', r['stats']['records_parsed'])
    print('main kernel %.3f ms for %.1f MB = %.1f GB/s (%.1f %% of 8 TB/s); step %.3f ms = %.2f G reads/s' % (
        ms, text.nbytes / 1e6, text.nbytes / ms / 1e6, text.nbytes / ms / 1e6 / 80.0, dt * 1e3, base_n * reps / dt / 1e9))
    print('three steps in flight: %.3f ms per step = %.2f G reads/s' % (dt3 * 1e3, base_n * reps / dt3 / 1e9))
    rl = r['stats']['readlengths']
    print('mean trimmed read length %.1f' % (sum(i * c for i, c in enumerate(rl)) / max(1, sum(rl))))


if __name__ == '__main__':
    main()
