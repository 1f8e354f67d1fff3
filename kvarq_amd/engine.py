"""
``kvarq.engine`` on an MI355X: the Python module surface of the reference's C
extension (csrc/workhorse.c:1204-1634) -- ``config``, ``get_config``,
``findseqs``, ``stats``, ``stop``, ``test`` and the ``Hit`` namedtuple -- over
the C ABI of libkvarq_hip.so (include/kvarq_hip.h) via ctypes.

Same names, argument meaning, return shapes and exceptions as the reference
(SURVEY.md section 8b).  Strings: file names and sequences may be ``str`` or
``bytes``; ``hitseqs`` come back in the type the sequences were given in
(``str`` via latin-1 for ``str`` input, like the reference under Python 2).

The scan itself runs only on the GPU: without the library or without a device
``findseqs`` raises, it never falls back to a CPU implementation.
"""
import collections
import itertools
import numpy as np
import ctypes as C

from . import _lib
from .fastq import FastqFileFormatException
from .log import lo

# workhorse.c:1579-1596
Hit = collections.namedtuple('Hit', 'seq_nr file_pos seq_pos length readlength')
Hit.__doc__ = (
    "seq_nr : refers to the list of sequences in call to engine.findseqs\n"
    "file_pos : beginning of read (within decompressed data)\n"
    "seq_pos : places the beginning of the read relative to the beginning of the sequence\n"
    "length : gives the number of overlapping basepairs\n"
    "readlength : length of the (quality trimmed) read containing the hit\n")

_KEYS = ('maxerrors', 'minoverlap', 'minreadlength', 'nthreads', 'Amin', 'Azero')


def _char(x, name):
    if isinstance(x, str):
        x = x.encode('latin-1')
    if not isinstance(x, (bytes, bytearray)) or len(x) != 1:
        raise TypeError('%s must be a single character' % name)       # format "c", workhorse.c:1503
    v = x[0]
    return v - 256 if v > 127 else v


def config(**kw):
    """config(**kwargs) -- configure the engine (workhorse.c:1498-1507): maxerrors,
    minoverlap, minreadlength, nthreads, Amin, Azero; values persist across calls"""
    for k in kw:
        if k not in _KEYS:
            raise TypeError("'%s' is an invalid keyword argument for this function" % k)
    L = _lib.lib()
    c = _lib.Config()
    L.kvq_config_get(C.byref(c))
    for k in _KEYS[:4]:
        if k in kw:
            if isinstance(kw[k], bool) or not isinstance(kw[k], int):
                raise TypeError('an integer is required')
            setattr(c, k, kw[k])
    for k in _KEYS[4:]:
        if k in kw:
            setattr(c, k, _char(kw[k], k))
    L.kvq_config_set(C.byref(c))


def get_config():
    """get_config() -- the current config as dictionary (workhorse.c:1484-1493)"""
    L = _lib.lib()
    c = _lib.Config()
    L.kvq_config_get(C.byref(c))
    return {'maxerrors': c.maxerrors, 'minoverlap': c.minoverlap, 'minreadlength': c.minreadlength,
            'nthreads': c.nthreads, 'Amin': chr(c.Amin & 0xFF), 'Azero': chr(c.Azero & 0xFF)}


class RescanRequired(MemoryError):
    """kvq_scan_finish after host batches: the hit arena was too small; it has been enlarged, and the same
    batches have to be fed again after a reset (``scan.Scanner`` does that itself)"""


def _raise_last():
    code, msg = _lib.last_error()
    if code == _lib.ERR_FORMAT:
        raise FastqFileFormatException(msg)
    if code == _lib.ERR_IO:
        raise IOError(msg)
    if code == _lib.ERR_RESCAN:
        raise RescanRequired(msg)
    if code == _lib.ERR_MEMORY:
        raise MemoryError(msg)
    if code == _lib.ERR_TYPE:
        raise TypeError(msg)
    raise RuntimeError(msg)


def _stats_dict(readlengths, longest, nseqbasehits, nseqhits, parsed, total, sigints, records):
    # workhorse.c:1205-1244; progress in float32 like the reference (1230-1232)
    progress = 0.0
    if total > 0:
        progress = C.c_float(C.c_float(min(parsed, total)).value / C.c_float(total).value).value
    return {
        'readlengths': tuple(readlengths[i] if i < _lib.MAX_READLENGTH else 0 for i in range(longest + 1)),
        'progress': progress,
        'nseqbasehits': tuple(nseqbasehits),
        'nseqhits': tuple(nseqhits),
        'parsed': parsed, 'total': total, 'sigints': sigints, 'records_parsed': records,
    }


def findseqs(fname, sequences):
    """findseqs(fname, sequences) -- finds occurences of base sequences in fastq files
    (workhorse.c:1249-1464).

    fname: file name or sequence of file names (plain or ``.gz``), scanned as one
    stream; sequences: sequence of strings.  Returns ``{'hits': tuple of Hit,
    'stats': dict as stats(), 'hitseqs': list of hit base strings}``."""
    import os, time
    t_in = time.perf_counter()
    L = _lib.lib()
    if isinstance(fname, (str, bytes)):
        fnames = [fname]
    else:
        try:
            fnames = list(fname)
        except TypeError:
            raise TypeError('fname must be [sequence of] string[s]')              # workhorse.c:1295
    for f in fnames:
        if not isinstance(f, (str, bytes)):
            raise TypeError('fname must be [sequence of] string[s]')
    try:
        seqs = list(sequences)
    except TypeError:
        raise TypeError('seqlist must be sequence of strings')                    # workhorse.c:1302
    as_str = any(isinstance(s, str) for s in seqs) or not seqs and isinstance(fnames[0] if fnames else '', str)
    bseqs = []
    for s in seqs:
        if isinstance(s, str):
            s = s.encode('latin-1')
        elif not isinstance(s, (bytes, bytearray)):
            raise TypeError('seqlist must be list of strings')                    # workhorse.c:1331
        bseqs.append(bytes(s))
    bfiles = [f.encode() if isinstance(f, str) else f for f in fnames]

    n = len(bseqs)
    farr = (C.c_char_p * max(1, len(bfiles)))(*bfiles)
    # sequences may hold NUL bytes: pass raw buffers, not c_char_p strings
    bufs = [C.create_string_buffer(s, len(s) + 1) for s in bseqs]
    sarr = (C.c_char_p * max(1, n))(*[C.cast(b, C.c_char_p) for b in bufs])
    lens = (C.c_int32 * max(1, n))(*[len(s) for s in bseqs])

    # ctypes releases the GIL for the duration of the call (workhorse.c:1377-1408)
    t_0 = time.perf_counter()
    h = L.kvq_findseqs(farr, len(bfiles), sarr, lens, n)
    t_1 = time.perf_counter()
    try:
        code, _ = _lib.last_error()
        if not h or code:
            _raise_last()
        nh = L.kvq_scan_n_hits(h)
        # (the arrays become Python objects in bulk: one ctypes index operation per field of every hit cost more
        # than the scan of a 3 GB file)
        def col(ptr, count):
            return np.ctypeslib.as_array(ptr, shape=(count,)).tolist() if count else []
        # (tuple.__new__(Hit, fields) is what Hit(*fields) ends up calling, without the Python-level frame in between)
        hits = tuple(map(tuple.__new__, itertools.repeat(Hit),
                         zip(col(L.kvq_scan_hit_seq_nr(h), nh), col(L.kvq_scan_hit_file_pos(h), nh), col(L.kvq_scan_hit_seq_pos(h), nh),
                             col(L.kvq_scan_hit_length(h), nh), col(L.kvq_scan_hit_readlength(h), nh))))
        off = col(L.kvq_scan_hitseq_offsets(h), nh + 1)
        blob = C.string_at(L.kvq_scan_hitseq_blob(h), off[nh]) if nh else b''
        if as_str:
            blob = blob.decode('latin-1')
        hitseqs = [blob[a:b] for a, b in zip(off, off[1:])]
        # counters layout (include/kvarq_hip.h): 4 scalars, readlengths[1024], nseqhits[n], nseqbasehits[n], ...
        o_hits = _lib.CTR_READLENGTHS + _lib.MAX_READLENGTH
        ctr = col(L.kvq_scan_counters(h), o_hits + 2 * n)
        st = _stats_dict(
            ctr[_lib.CTR_READLENGTHS:o_hits], ctr[_lib.CTR_LONGEST] - 1,
            ctr[o_hits + n:o_hits + 2 * n], ctr[o_hits:o_hits + n],
            L.kvq_scan_parsed(h), L.kvq_scan_total(h), _sigints(), ctr[_lib.CTR_RECORDS])
        if os.environ.get('KVQ_TIMING'):
            import sys
            sys.stderr.write('engine.findseqs: arguments %.1f ms, library %.1f ms, results as Python objects %.1f ms\n' % ((t_0 - t_in) * 1e3, (t_1 - t_0) * 1e3, (time.perf_counter() - t_1) * 1e3))
        return {'hits': hits, 'stats': st, 'hitseqs': hitseqs}
    finally:
        if h:
            t_f = time.perf_counter()
            L.kvq_findseqs_free(h)
            if os.environ.get('KVQ_TIMING'):
                import sys
                sys.stderr.write('engine.findseqs: free %.1f ms\n' % ((time.perf_counter() - t_f) * 1e3))


def _sigints():
    L = _lib.lib()
    ls = _lib.LiveStats()
    L.kvq_poll_stats(C.byref(ls), None, None, None, 0)
    return ls.sigints


def stats():
    """stats() -- statistics of the running (or last) scan (workhorse.c:1205-1244);
    may be called from another thread while findseqs runs"""
    L = _lib.lib()
    ls = _lib.LiveStats()
    L.kvq_poll_stats(C.byref(ls), None, None, None, 0)
    n = ls.nseq
    rls = (C.c_int64 * _lib.MAX_READLENGTH)()
    sh = (C.c_int64 * max(1, n))()
    sbh = (C.c_int64 * max(1, n))()
    L.kvq_poll_stats(C.byref(ls), rls, sh, sbh, n)
    return _stats_dict(list(rls), ls.rls_longest, list(sbh)[:n], list(sh)[:n], ls.parsed, ls.total, ls.sigints,
                       ls.records_parsed)


def stop():
    """stop() -- stops the scanning process; findseqs returns the hits found so far
    (workhorse.c:1469-1479)"""
    lo.debug('engine stopped')
    _lib.lib().kvq_request_stop()


def test():
    """test() -- no-op (the reference dumps its disabled profiler, workhorse.c:1512-1517)"""
    return None


def install_sigint_counter():
    """the reference installs, at import, a C SIGINT handler that only counts
    (workhorse.c:133-136, 1632; the CLI turns two of them into stop(), cli.py:156-164).
    Replacing the process's handler is a process-wide side effect, so here it is an
    explicit call: the ``kvarq/engine.py`` shim of INTEGRATION.md makes it.  The handler
    lives in the library (``kvq_sigint_counter_install``): it counts while the main
    thread is inside ``findseqs``, where a Python-level handler would have to wait."""
    if _lib.lib().kvq_sigint_counter_install():
        _raise_last()


def remove_sigint_counter():
    """puts back the handler that was there before ``install_sigint_counter``"""
    _lib.lib().kvq_sigint_counter_remove()
