#!/usr/bin/env python3
"""What the host side of the file path can reach on this box: H2D from pinned memory alone, page-cache reads
(pread into pinned memory, T threads) alone, and both at once.  usage: python tools/r3_h2d.py [MB per batch] [threads ...]"""
import os, sys, time, threading
import numpy as np, torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ths = [int(x) for x in sys.argv[2:]] or [8, 16, 32]
n = mb << 20; nb = 48
path = '/tmp/kvq_h2d.bin'
np.random.default_rng(1).integers(0, 255, n * 8, dtype=np.uint8).tofile(path)        # 8 batches of file, read round-robin
fd = os.open(path, os.O_RDONLY)
pins = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
devs = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(2)]
cs = torch.cuda.Stream()
def h2d_only():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        for i in range(nb): devs[i & 1].copy_(pins[i & 1], non_blocking=True)
    cs.synchronize(); return time.perf_counter() - t0
def read_into(buf, off, T):
    mv = memoryview(buf.numpy()); per = (n + T - 1) // T; per = (per + 4095) & ~4095
    def work(k):
        a = k * per; b = min(n, a + per)
        while a < b:
            got = os.preadv(fd, [mv[a:b]], off + a); a += got
    ts = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in ts: t.start()
    for t in ts: t.join()
def read_only(T):
    t0 = time.perf_counter()
    for i in range(nb): read_into(pins[i & 1], (i % 8) * n, T)
    return time.perf_counter() - t0
def both(T):
    evs = [None, None]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(nb):
        if evs[i & 1] is not None: evs[i & 1].synchronize()
        read_into(pins[i & 1], (i % 8) * n, T)
        with torch.cuda.stream(cs):
            devs[i & 1].copy_(pins[i & 1], non_blocking=True)
            e = torch.cuda.Event(); e.record(cs); evs[i & 1] = e
    cs.synchronize(); return time.perf_counter() - t0
h2d_only(); gb = nb * n / 1e9
print('H2D alone (pinned, %d MB batches): %.1f GB/s' % (mb, gb / min(h2d_only() for _ in range(3))))
for T in ths:
    read_only(T)
    print('T=%2d  pread alone %.1f GB/s   pread + H2D overlapped %.1f GB/s' % (T, gb / min(read_only(T) for _ in range(3)), gb / min(both(T) for _ in range(3))))
os.close(fd); os.remove(path)
