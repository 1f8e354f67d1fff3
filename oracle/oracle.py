"""
oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of libkvarq_oracle.so (this repo's CPU restatement of
/root/reference/csrc/workhorse.c) plus a loader for the reference's own engine
compiled into oracle/_ref (where it was built).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by kvarq_amd.

Results are returned in the shape of ``kvarq.engine.findseqs`` under CPython 3
(workhorse.c:1434-1437): ``{'hits': tuple[Hit], 'stats': {...}, 'hitseqs': [bytes]}``.
"""
import collections
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libkvarq_oracle.so')

Hit = collections.namedtuple('Hit', 'seq_nr file_pos seq_pos length readlength')

MAX_READLENGTH = 1024

ERR_FORMAT, ERR_IO, ERR_MEMORY, ERR_RUNTIME = 1, 2, 3, 4


class OracleFormatError(Exception):
    """stands for kvarq.fastq.FastqFileFormatException"""


class Config(C.Structure):
    _fields_ = [('maxerrors', C.c_int32), ('minoverlap', C.c_int32),
                ('minreadlength', C.c_int32), ('nthreads', C.c_int32),
                ('Amin', C.c_int8), ('Azero', C.c_int8)]


class Result(C.Structure):
    _fields_ = [('n_hits', C.c_int64),
                ('seq_nr', C.POINTER(C.c_int32)), ('file_pos', C.POINTER(C.c_int64)),
                ('seq_pos', C.POINTER(C.c_int32)), ('length', C.POINTER(C.c_int32)),
                ('readlength', C.POINTER(C.c_int32)),
                ('hitseq_blob', C.POINTER(C.c_uint8)), ('hitseq_off', C.POINTER(C.c_int64)),
                ('readlengths', C.c_int64 * MAX_READLENGTH), ('rls_longest', C.c_int64),
                ('nseq', C.c_int32),
                ('nseqhits', C.POINTER(C.c_int64)), ('nseqbasehits', C.POINTER(C.c_int64)),
                ('records_parsed', C.c_int64), ('parsed', C.c_int64), ('total', C.c_int64),
                ('err_code', C.c_int32), ('errmsg', C.c_char * 1024),
                ('cap_hits', C.c_int64), ('cap_blob', C.c_int64)]


_lib = None


def build():
    """compile the restatement (and, where /root/reference exists, oracle/_ref)"""
    subprocess.check_call(['make', '-s', '-C', HERE, 'libkvarq_oracle.so'])
    if os.path.isdir(os.environ.get('KVARQ_REFERENCE', '/root/reference')):
        subprocess.check_call(['make', '-s', '-C', HERE, 'ref'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.kvo_findseqs.restype = C.POINTER(Result)
        L.kvo_findseqs.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p),
                                   C.POINTER(C.c_int32), C.c_int, C.POINTER(Config)]
        L.kvo_scan_memory.restype = C.POINTER(Result)
        L.kvo_scan_memory.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_int32), C.c_int, C.POINTER(Config)]
        L.kvo_free.argtypes = [C.POINTER(Result)]
        L.kvo_chunk_offsets.restype = C.c_int64
        L.kvo_chunk_offsets.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int64]
        L.kvo_fold_coverage.argtypes = [C.POINTER(Result), C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _b(x):
    return x if isinstance(x, bytes) else x.encode('latin-1')


def _cfg(maxerrors=0, minoverlap=20, minreadlength=10, nthreads=1, Amin=b'!', Azero=b'!'):
    # defaults = the reference's module globals, workhorse.c:71-75
    return Config(maxerrors, minoverlap, minreadlength, nthreads, _b(Amin)[0], _b(Azero)[0])


def _seq_args(seqs):
    seqs = [_b(s) for s in seqs]
    arr = (C.c_char_p * max(1, len(seqs)))(*seqs)
    lens = (C.c_int32 * max(1, len(seqs)))(*[len(s) for s in seqs])
    return seqs, arr, lens


def _raise(code, msg):
    if code == ERR_FORMAT:
        raise OracleFormatError(msg)
    if code == ERR_IO:
        raise IOError(msg)
    if code == ERR_MEMORY:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def _convert(rp, seqs=None, fold=False):
    L = lib()
    try:
        if not rp:
            raise MemoryError('oracle out of memory')
        r = rp.contents
        if r.err_code:
            _raise(r.err_code, r.errmsg.decode('latin-1'))
        n = r.n_hits
        hits = tuple(Hit(r.seq_nr[i], r.file_pos[i], r.seq_pos[i], r.length[i], r.readlength[i])
                     for i in range(n))
        blob = C.string_at(r.hitseq_blob, r.hitseq_off[n]) if n else b''
        hitseqs = [blob[r.hitseq_off[i]:r.hitseq_off[i + 1]] for i in range(n)]
        total = r.total
        stats = {
            # entries >= MAX_READLENGTH are out-of-bounds reads in the reference
            # (workhorse.c:1214-1216 vs 107); they are zeros here
            'readlengths': tuple(r.readlengths[i] if i < MAX_READLENGTH else 0 for i in range(r.rls_longest + 1)),
            # float32 arithmetic as in workhorse.c:1230-1232
            'progress': (C.c_float(C.c_float(min(r.parsed, total)).value / C.c_float(total).value).value
                         if total > 0 else 0.0),
            'nseqbasehits': tuple(r.nseqbasehits[i] for i in range(r.nseq)),
            'nseqhits': tuple(r.nseqhits[i] for i in range(r.nseq)),
            'parsed': r.parsed, 'total': total, 'sigints': 0,
            'records_parsed': r.records_parsed,
        }
        out = {'hits': hits, 'stats': stats, 'hitseqs': hitseqs}
        if fold:
            bs, arr, lens = _seq_args(seqs)
            off = [0]
            for s in bs:
                off.append(off[-1] + len(s))
            offc = (C.c_int64 * len(off))(*off)
            cov = (C.c_int64 * max(1, off[-1]))()
            mut = (C.c_int64 * max(1, off[-1] * 6))()
            L.kvo_fold_coverage(rp, arr, lens, offc, cov, mut)
            out['coverage'] = list(cov)[:off[-1]]
            out['mutations'] = list(mut)[:off[-1] * 6]
        return out
    finally:
        if rp:
            L.kvo_free(rp)


def findseqs(fname, seqs, fold=False, **config):
    """engine.findseqs(fname, seqs) with the engine config given as keywords"""
    L = lib()
    fnames = [fname] if isinstance(fname, (str, bytes)) else list(fname)
    fnames = [_b(f) for f in fnames]
    farr = (C.c_char_p * max(1, len(fnames)))(*fnames)
    bs, arr, lens = _seq_args(seqs)
    cfg = _cfg(**config)
    rp = L.kvo_findseqs(farr, len(fnames), arr, lens, len(bs), C.byref(cfg))
    return _convert(rp, bs, fold)


def scan_memory(data, seqs, fpos_base=0, fold=False, **config):
    """scan an in-memory FastQ stream (bytes / bytearray / numpy uint8 array)"""
    L = lib()
    bs, arr, lens = _seq_args(seqs)
    cfg = _cfg(**config)
    if hasattr(data, 'ctypes'):
        ptr, n = data.ctypes.data, data.nbytes
    else:
        buf = (C.c_char * len(data)).from_buffer_copy(bytes(data)) if len(data) else (C.c_char * 1)()
        ptr, n = C.addressof(buf), len(data)
    rp = L.kvo_scan_memory(ptr, n, fpos_base, arr, lens, len(bs), C.byref(cfg))
    return _convert(rp, bs, fold)


def chunk_offsets(data):
    L = lib()
    if hasattr(data, 'ctypes'):
        ptr, n = data.ctypes.data, data.nbytes
    else:
        buf = (C.c_char * max(1, len(data))).from_buffer_copy(bytes(data) or b'\0')
        ptr, n = C.addressof(buf), len(data)
    cap = n // (512 * 1024) + 4
    out = (C.c_int64 * (cap + 1))()
    k = L.kvo_chunk_offsets(ptr, n, out, cap)
    if k < 0:
        raise RuntimeError('could find beginning of record')
    return list(out)[:k + 1]


# ---------------------------------------------------------------------------
# the reference's own engine (oracle/_ref), where it has been built
# ---------------------------------------------------------------------------

_ref_engine = None
_ref_modules = {}


def ref_engine():
    """the reference C engine compiled by oracle/build_ref.sh, or None"""
    global _ref_engine
    if _ref_engine is None:
        d = os.path.join(HERE, '_ref')
        if not os.path.exists(os.path.join(d, 'kvarq', 'engine.so')):
            return None
        saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'kvarq' or k.startswith('kvarq.')}
        sys.path.insert(0, d)
        try:
            from kvarq import engine  # noqa
            _ref_engine = engine
        except Exception:
            _ref_engine = False
        finally:
            sys.path.remove(d)
            # the stub package the extension resolved its collaborators from stays alive here (the extension holds the objects it
            # looked up at import), but NOT under the name `kvarq`: that name belongs to whatever had it before -- the product's
            # own kvarq/ shim package when a test imports it -- so `from kvarq import engine` never means the reference build
            global _ref_modules
            _ref_modules = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'kvarq' or k.startswith('kvarq.')}
            sys.modules.update(saved)
    return _ref_engine or None


def ref_findseqs(fname, seqs, **config):
    """run the reference engine; bytes in, reference-shaped dict out.
    Never pass a missing file (use-after-free in the reference, workhorse.c:664-667)."""
    eng = ref_engine()
    if eng is None:
        raise RuntimeError('oracle/_ref is not built')
    fnames = _b(fname) if isinstance(fname, (str, bytes)) else tuple(_b(f) for f in fname)
    for f in ([fnames] if isinstance(fnames, bytes) else fnames):
        if not os.path.exists(f):
            raise IOError('refusing to hand a missing file to the reference engine')
    cfg = dict(maxerrors=0, minoverlap=20, minreadlength=10, nthreads=1, Amin=b'!', Azero=b'!')
    cfg.update(config)
    cfg['Amin'] = _b(cfg['Amin'])
    cfg['Azero'] = _b(cfg['Azero'])
    eng.config(**cfg)
    return eng.findseqs(fnames, [_b(s) for s in seqs])
