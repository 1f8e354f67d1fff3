"""
Object view of the scan C ABI (include/kvarq_hip.h) for callers that keep data
resident on the GPU: ``Table`` (the sequence list of ``engine.findseqs`` with
its seed index) and ``Scanner`` (``scan_filepart`` of the reference,
csrc/workhorse.c:976-1197, over batches that are already in device or host
memory).  ``engine.findseqs`` is the file-level entry; bench.py and the
multi-GPU driver use this module.
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import Hit, _raise_last


def _check(rc):
    if rc:
        _raise_last()


class Table(object):
    """target sequences (bytes or str) + engine config -> device table"""

    def __init__(self, seqs, **config):
        L = _lib.lib()
        self.seqs = [s.encode('latin-1') if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(self.seqs)
        cfg = _lib.Config()
        L.kvq_config_get(C.byref(cfg))
        for k, v in config.items():
            if k in ('Amin', 'Azero'):
                v = (v.encode('latin-1') if isinstance(v, str) else v)[0]
                v = v - 256 if v > 127 else v
            setattr(cfg, k, v)
        self.config = cfg
        bufs = [C.create_string_buffer(s, len(s) + 1) for s in self.seqs]
        arr = (C.c_char_p * max(1, n))(*[C.cast(b, C.c_char_p) for b in bufs])
        lens = (C.c_int32 * max(1, n))(*[len(s) for s in self.seqs])
        self.h = L.kvq_table_create(arr, lens, n, C.byref(cfg))
        if not self.h:
            _raise_last()
        self.nseq = n
        self.bases = L.kvq_table_bases(self.h)
        self.counters_len = L.kvq_counters_len(self.h)
        self.off_nseqhits = L.kvq_counters_off_nseqhits(self.h)
        self.off_nseqbasehits = L.kvq_counters_off_nseqbasehits(self.h)
        self.off_coverage = L.kvq_counters_off_coverage(self.h)
        self.off_mutations = L.kvq_counters_off_mutations(self.h)
        self.seq_offset = [L.kvq_table_seq_offset(self.h, i) for i in range(n + 1)]
        self.seeded = [bool(L.kvq_table_seq_is_seeded(self.h, i)) for i in range(n)]
        self.seed_k = L.kvq_table_seed_k(self.h)

    def close(self):
        if self.h:
            _lib.lib().kvq_table_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chunk_offsets(data):
    """chunk cuts of fastq_read (workhorse.c:737-956) on an in-memory stream (numpy uint8 / bytes)"""
    L = _lib.lib()
    arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    cap = arr.nbytes // (512 * 1024) + 4
    out = (C.c_int64 * (cap + 1))()
    n = L.kvq_chunk_offsets(arr.ctypes.data if arr.nbytes else None, arr.nbytes, out, cap)
    if n < 0:
        raise RuntimeError('could find beginning of record')
    return np.array(list(out)[:n + 1], dtype=np.int64)


def _view(ptr, n, ctype, dtype):
    """numpy view of a C array the library owns (np.ctypeslib.as_array costs ~60 us per call)"""
    addr = C.cast(ptr, C.c_void_p).value
    if not addr or n <= 0:
        return np.zeros(0, dtype=dtype)
    return np.frombuffer((ctype * n).from_address(addr), dtype=dtype)


class RescanRequired(RuntimeError):
    """the hit arena overflowed on host batches that the Scanner no longer holds: reset() and feed them again"""


class Scanner(object):
    """accumulates hits and counters over any number of batches.

    Host batches (``scan_host``) cannot be replayed by the library when the hit arena turns out too small
    (``KVQ_ERR_RESCAN``); ``finish`` feeds them again itself from COPIES it keeps -- up to `retain_limit` bytes
    (the caller's buffers are never kept: they may be refilled).  Beyond that limit, or with ``retain_limit=0``,
    ``finish`` raises ``RescanRequired`` and the caller, who has the data, resets and feeds again; a caller with
    a re-readable source passes ``replay=callable`` (called with the scanner after ``reset()``) instead."""

    def __init__(self, table, counters_ptr=None, retain_limit=1 << 30, replay=None):
        self.table = table
        self.h = _lib.lib().kvq_scan_create(table.h, counters_ptr)
        if not self.h:
            _raise_last()
        self._host_batches = []          # copies of what scan_host was given since the last reset (fed again when the hit arena overflows)
        self._device_batches = []        # what scan_device was given since the last reset (the caller's device memory: nothing is copied)
        self._retained, self._retain_limit, self._replay = 0, (0 if replay is not None else retain_limit), replay
        self._comm = None

    def set_comm(self, comm):
        """several GPUs (``dist.NativeComm``): ``finish`` becomes collective and sums the counters of all ranks over RCCL"""
        self._comm = comm
        _check(_lib.lib().kvq_scan_set_comm(self.h, comm.h if comm is not None else None))

    def gather_hits(self):
        """collective, after ``finish(hits=False)``: every rank's hits, in rank (= stream) order, on every rank;
        returns what ``finish`` returns for them"""
        _check(_lib.lib().kvq_scan_gather_hits(self.h, self._comm.h if self._comm is not None else None))
        return self._hits_dict()

    def force_exhaustive(self, on=True):
        _lib.lib().kvq_scan_force_exhaustive(self.h, 1 if on else 0)

    def scan_device(self, d_ptr, nbytes, chunk_off, fpos_base=0):
        co = np.ascontiguousarray(chunk_off, dtype=np.int64)
        self._device_batches.append((d_ptr, nbytes, co, fpos_base))
        _check(_lib.lib().kvq_scan_device(self.h, d_ptr, nbytes, co.ctypes.data_as(C.POINTER(C.c_int64)), len(co) - 1, fpos_base))

    def _feed_host(self, arr, co, fpos_base):
        _check(_lib.lib().kvq_scan_host(self.h, arr.ctypes.data if arr.nbytes else None, arr.nbytes,
                                        co.ctypes.data_as(C.POINTER(C.c_int64)), len(co) - 1, fpos_base))

    def scan_host(self, data, chunk_off=None, fpos_base=0):
        arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        co = chunk_offsets(arr) if chunk_off is None else np.ascontiguousarray(chunk_off, dtype=np.int64)
        if self._host_batches is not None:
            if self._retained + arr.nbytes <= self._retain_limit:
                self._host_batches.append((arr.copy(), co.copy(), fpos_base)); self._retained += arr.nbytes
            else:
                self._host_batches = None          # too much to keep: an overflow is the caller's to replay
        self._feed_host(arr, co, fpos_base)

    def finish_begin(self):
        """every batch of this scan has been fed: enqueue the ordering of the hits and the copies to the host now
        (``kvq_scan_finish_begin``); ``finish`` then only waits.  For callers with several scanners in flight."""
        _check(_lib.lib().kvq_scan_finish_begin(self.h))

    def finish(self, hits=True, stats=True):
        """-> dict with 'hits', 'hitseqs' (bytes), 'stats', 'coverage', 'mutations', 'counters'.

        ``hits=False`` leaves hits and hit bytes in the library's arrays (``kvq_scan_hit_*``), ``stats=False``
        also skips the Python ``stats`` dict and hands out ``counters`` as a view of the library's host
        array (valid until the next ``reset`` / ``close``) -- for callers that only reduce counters."""
        L = _lib.lib()
        for attempt in range(4):
            if L.kvq_scan_finish(self.h) == 0:
                break
            if _lib.last_error()[0] != _lib.ERR_RESCAN or attempt == 3:
                _raise_last()
            # the hit arena was too small for the host batches (it has been enlarged): feed them again
            # (with a communicator the overflow may be another rank's: every rank goes round again, whatever it was fed with)
            again, device, replay = self._host_batches, self._device_batches, self._replay
            if again is None and replay is None:
                raise RescanRequired(_lib.last_error()[1])
            if replay is None and not again and not device:
                raise RescanRequired('nothing to feed again: ' + str(_lib.last_error()[1]))
            retained = self._retained
            self.reset()
            if replay is not None:
                replay(self)
            else:
                # the kept copies go in again as they are (no second copy, and they stay kept: a further round may want them);
                # device batches are the caller's memory and are simply named again
                self._host_batches, self._retained = again, retained
                for arr, co, fpos_base in again:
                    self._feed_host(arr, co, fpos_base)
                for d_ptr, nbytes, co, fpos_base in device:
                    self.scan_device(d_ptr, nbytes, co, fpos_base)
        t = self.table
        ctr = _view(L.kvq_scan_counters(self.h), t.counters_len, C.c_int64, np.int64)
        if stats:
            ctr = ctr.copy()
        out = {'counters': ctr}
        out['n_hits'] = L.kvq_scan_n_hits(self.h)
        if hits:
            out.update(self._hits_dict())
        out['kernel_ms'] = L.kvq_scan_kernel_ms(self.h)
        out['main_kernel_ms'] = L.kvq_scan_main_kernel_ms(self.h)
        out['main_kernel_launches'] = L.kvq_scan_main_kernel_launches(self.h)
        if not stats:
            return out
        longest = int(ctr[_lib.CTR_LONGEST]) - 1
        parsed, total = L.kvq_scan_parsed(self.h), L.kvq_scan_total(self.h)
        rls = ctr[_lib.CTR_READLENGTHS:_lib.CTR_READLENGTHS + _lib.MAX_READLENGTH]
        out['stats'] = {
            'readlengths': tuple(int(rls[i]) if i < _lib.MAX_READLENGTH else 0 for i in range(longest + 1)),
            'progress': (C.c_float(C.c_float(min(parsed, total)).value / C.c_float(total).value).value if total > 0 else 0.0),
            'nseqbasehits': tuple(int(x) for x in ctr[t.off_nseqbasehits:t.off_nseqbasehits + t.nseq]),
            'nseqhits': tuple(int(x) for x in ctr[t.off_nseqhits:t.off_nseqhits + t.nseq]),
            'parsed': parsed, 'total': total, 'sigints': 0, 'records_parsed': int(ctr[_lib.CTR_RECORDS]),
        }
        out['coverage'] = ctr[t.off_coverage:t.off_coverage + t.bases]
        out['mutations'] = ctr[t.off_mutations:t.off_mutations + 6 * t.bases]
        path = L.kvq_scan_path(self.h)
        out['path'] = {'seeded': bool(path & 1), 'exhaustive': bool(path & 2), 'rescanned': bool(path & 4), 'tiles_rescanned': bool(path & 8)}
        return out

    def hit_arrays(self):
        """the library's result arrays as numpy arrays (copies): seq_nr, file_pos, seq_pos, length, readlength,
        hitseq offsets (n + 1), hit bytes"""
        L = _lib.lib()
        nh = L.kvq_scan_n_hits(self.h)
        a = [_view(f(self.h), nh, ct, dt).copy()
             for f, ct, dt in ((L.kvq_scan_hit_seq_nr, C.c_int32, np.int32), (L.kvq_scan_hit_file_pos, C.c_int64, np.int64),
                               (L.kvq_scan_hit_seq_pos, C.c_int32, np.int32), (L.kvq_scan_hit_length, C.c_int32, np.int32),
                               (L.kvq_scan_hit_readlength, C.c_int32, np.int32))]
        off = _view(L.kvq_scan_hitseq_offsets(self.h), nh + 1, C.c_int64, np.int64).copy()
        blob = np.frombuffer(C.string_at(L.kvq_scan_hitseq_blob(self.h), int(off[nh])) if nh else b'', dtype=np.uint8)
        return dict(seq_nr=a[0], file_pos=a[1], seq_pos=a[2], length=a[3], readlength=a[4], offsets=off, blob=blob)

    def _hits_dict(self):
        """the library's result arrays as the reference's ``hits`` tuple and ``hitseqs`` list"""
        L = _lib.lib()
        nh = L.kvq_scan_n_hits(self.h)
        a = [_view(f(self.h), nh, ct, dt).copy()
             for f, ct, dt in ((L.kvq_scan_hit_seq_nr, C.c_int32, np.int32), (L.kvq_scan_hit_file_pos, C.c_int64, np.int64),
                               (L.kvq_scan_hit_seq_pos, C.c_int32, np.int32), (L.kvq_scan_hit_length, C.c_int32, np.int32),
                               (L.kvq_scan_hit_readlength, C.c_int32, np.int32))]
        hits = tuple(Hit(int(a[0][i]), int(a[1][i]), int(a[2][i]), int(a[3][i]), int(a[4][i])) for i in range(nh))
        off = L.kvq_scan_hitseq_offsets(self.h)
        blob = C.string_at(L.kvq_scan_hitseq_blob(self.h), off[nh]) if nh else b''
        return {'n_hits': nh, 'hits': hits, 'hitseqs': [blob[off[i]:off[i + 1]] for i in range(nh)]}

    def reset(self):
        self._host_batches, self._device_batches, self._retained = [], [], 0
        _check(_lib.lib().kvq_scan_reset(self.h))

    def close(self):
        if self.h:
            _lib.lib().kvq_scan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer(object):
    """a plain device allocation (hipMalloc) with host copies in and out"""

    def __init__(self, nbytes):
        self.nbytes = nbytes
        self.ptr = _lib.lib().kvq_device_alloc(nbytes + 64)
        if not self.ptr:
            _raise_last()

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        _check(_lib.lib().kvq_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes))

    def download(self, nbytes=None, offset=0):
        n = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(n, dtype=np.uint8)
        _check(_lib.lib().kvq_memcpy_d2h(out.ctypes.data, self.ptr + offset, n))
        return out

    def free(self):
        if self.ptr:
            _lib.lib().kvq_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
