#!/usr/bin/env python3
"""tools/r3_h2d.py with the pinned buffers allocated by hipHostMalloc under different flags (default, NumaUser, WriteCombined,
NonCoherent): does any of them lift the 40 GB/s that pread + H2D reach together?  usage: python tools/r3_h2d_flags.py [MB] [threads]"""
import ctypes as C, os, sys, time, threading
import numpy as np
hip = C.CDLL('libamdhip64.so')
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = mb << 20; nb = 48
path = '/tmp/kvq_h2d.bin'
np.random.default_rng(1).integers(0, 255, n * 8, dtype=np.uint8).tofile(path)
fd = os.open(path, os.O_RDONLY)
def chk(rc, what):
    if rc != 0: raise RuntimeError('%s -> %d' % (what, rc))
devs = []
for _ in range(2):
    p = C.c_void_p(); chk(hip.hipMalloc(C.byref(p), C.c_size_t(n)), 'hipMalloc'); devs.append(p)
stream = C.c_void_p(); chk(hip.hipStreamCreate(C.byref(stream)), 'stream')
evs = []
for _ in range(2):
    e = C.c_void_p(); chk(hip.hipEventCreate(C.byref(e)), 'event'); evs.append(e)
FLAGS = [('default', 0), ('NumaUser', 0x20000000), ('WriteCombined', 0x4), ('NonCoherent', 0x80000000), ('Coherent', 0x40000000)]
for name, fl in FLAGS:
    pins = []
    ok = True
    for _ in range(2):
        p = C.c_void_p()
        if hip.hipHostMalloc(C.byref(p), C.c_size_t(n), C.c_uint(fl)) != 0: ok = False; break
        pins.append(p)
    if not ok: print('%-14s hipHostMalloc refused' % name); continue
    views = [np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)) for p in pins]
    for v in views: v[::4096] = 1                                   # first touch by this thread
    def h2d_only():
        hip.hipDeviceSynchronize(); t0 = time.perf_counter()
        for i in range(nb): hip.hipMemcpyAsync(devs[i & 1], pins[i & 1], C.c_size_t(n), 1, stream)
        hip.hipStreamSynchronize(stream); return time.perf_counter() - t0
    def read_into(k, off):
        mv = memoryview(views[k]); per = ((n + T - 1) // T + 4095) & ~4095
        def work(j):
            a = j * per; b = min(n, a + per)
            while a < b: a += os.preadv(fd, [mv[a:b]], off + a)
        ts = [threading.Thread(target=work, args=(j,)) for j in range(T)]
        for t in ts: t.start()
        for t in ts: t.join()
    def read_only():
        t0 = time.perf_counter()
        for i in range(nb): read_into(i & 1, (i % 8) * n)
        return time.perf_counter() - t0
    def both():
        used = [False, False]
        hip.hipDeviceSynchronize(); t0 = time.perf_counter()
        for i in range(nb):
            if used[i & 1]: hip.hipEventSynchronize(evs[i & 1])
            read_into(i & 1, (i % 8) * n)
            hip.hipMemcpyAsync(devs[i & 1], pins[i & 1], C.c_size_t(n), 1, stream)
            hip.hipEventRecord(evs[i & 1], stream); used[i & 1] = True
        hip.hipStreamSynchronize(stream); return time.perf_counter() - t0
    gb = nb * n / 1e9
    h2d_only(); read_only()
    print('%-14s H2D alone %.1f GB/s   pread alone (T=%d) %.1f GB/s   both at once %.1f GB/s' % (
        name, gb / min(h2d_only() for _ in range(3)), T, gb / min(read_only() for _ in range(3)), gb / min(both() for _ in range(3))))
    for p in pins: hip.hipHostFree(p)
os.close(fd); os.remove(path)
