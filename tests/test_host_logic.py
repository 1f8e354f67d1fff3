"""
CPU-only checks of the product's host side: the C-ABI library loads and exports
every symbol include/kvarq_hip.h declares, the reader/chunker cuts the same
buffers fastq_read would (against the oracle's restatement), the synthetic
generator matches its numpy statement, and the engine module keeps the
reference's config semantics.  No compute entry point is called here.
"""
import ctypes as C
import gzip
import os
import re

import numpy as np
import pytest

import cases
from kvarq_amd import _lib, engine, synth
from kvarq_amd.fastq import FastqFileFormatException, Q2A
from oracle import oracle as O
from util import expected

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, 'include', 'kvarq_hip.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    declared = set(re.findall(r'\b(kvq_\w+)\s*\(', text))
    assert len(declared) > 40
    L = C.CDLL(_lib.PATH)
    for name in sorted(declared):
        assert hasattr(L, name), 'libkvarq_hip.so lacks %s' % name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    _lib.lib()


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    L = _lib.lib()
    if L.kvq_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(RuntimeError) as ei:
        engine.findseqs(os.path.join(cases.FASTQS, 'test_engine.fastq'), ['ACGT'])
    assert 'no HIP device' in str(ei.value)


def test_config_semantics():
    # workhorse.c:1484-1507, test_engine.py:134-139 (values persist across calls)
    saved = engine.get_config()
    try:
        engine.config(nthreads=1, maxerrors=2, minoverlap=25, Amin='!', Azero='!')
        engine.config(maxerrors=0)
        c = engine.get_config()
        assert c['maxerrors'] == 0 and c['minoverlap'] == 25 and c['Amin'] == '!'
        assert set(c) == {'maxerrors', 'minoverlap', 'minreadlength', 'nthreads', 'Amin', 'Azero'}
        engine.config(Amin=b'.')
        assert engine.get_config()['Amin'] == '.'
        with pytest.raises(TypeError):
            engine.config(bogus=1)
        with pytest.raises(TypeError):
            engine.config(Amin='ab')
        with pytest.raises(TypeError):
            engine.config(maxerrors='2')
        assert engine.config() is None and engine.test() is None
    finally:
        engine.config(**saved)
    assert engine.Hit._fields == ('seq_nr', 'file_pos', 'seq_pos', 'length', 'readlength')   # workhorse.c:1579-1586
    assert issubclass(FastqFileFormatException, Exception)
    assert Q2A(13) == '.'          # kvarq/config.py:3 with kvarq/fastq.py:245-247


def chunk_offsets(data):
    L = _lib.lib()
    arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    cap = arr.nbytes // (512 * 1024) + 4
    out = (C.c_int64 * (cap + 1))()
    n = L.kvq_chunk_offsets(arr.ctypes.data if arr.nbytes else None, arr.nbytes, out, cap)
    return n, list(out)[:max(n, 0) + 1]


def test_chunk_offsets_match_the_oracle():
    for data in (cases.multichunk(), cases.ragged(21, 7000, cases.RAGGED_TARGETS, nl='\r\n'), b'',
                 cases.quirk_probe(), b'@a\nA\n+\nI\n' * 300000):
        n, off = chunk_offsets(data)
        assert off == O.chunk_offsets(data), len(data)
        assert n == len(off) - 1
    # exact multiple of the buffer size: the cut leftover becomes one more chunk (workhorse.c:905-910)
    rec = b'@rr\n' + b'A' * 1020 + b'\n+\n' + b'I' * 1020 + b'\n'         # 2048 bytes
    data = rec * 1024
    assert len(data) == 2 * 1024 * 1024
    n, off = chunk_offsets(data)
    assert off == O.chunk_offsets(data)
    assert n == 3 and off[1] == 1024 * 1024 - 2048


def plan(files, batch=0):
    L = _lib.lib()
    farr = (C.c_char_p * len(files))(*[f.encode() for f in files])
    cap = 1 << 16
    fpos, ln = (C.c_int64 * cap)(), (C.c_int64 * cap)()
    parsed, total = C.c_int64(), C.c_int64()
    n = L.kvq_host_chunk_plan(farr, len(files), fpos, ln, cap, C.byref(parsed), C.byref(total), batch)
    if n < 0:
        return None, _lib.last_error()
    return [(fpos[i], ln[i]) for i in range(n)], (parsed.value, total.value)


def expected_plan(streams):
    out, base = [], 0
    for s in streams:
        off = O.chunk_offsets(s)
        out += [(base + a, b - a) for a, b in zip(off[:-1], off[1:])]
        base += len(s)
    return out


@pytest.mark.parametrize('batch', [0, 1 << 20, 3 << 20])
def test_reader_cuts_the_chunks_fastq_read_would(tmp_path, batch):
    big = cases.multichunk()
    other = cases.ragged(33, 5000, cases.RAGGED_TARGETS)
    p1, p2, p3 = str(tmp_path / 'a.fastq'), str(tmp_path / 'b.fastq.gz'), str(tmp_path / 'c.fastq')
    open(p1, 'wb').write(big)
    with open(p2, 'wb') as f:                      # two gzip members in one file (workhorse.c:842-866)
        f.write(gzip.compress(other[:len(other) // 2], mtime=0))
        f.write(gzip.compress(other[len(other) // 2:], mtime=0))
    open(p3, 'wb').write(b'@tail\nACGT')           # partial record only
    got, (parsed, total) = plan([p1, p2, p3], batch)
    assert got == expected_plan([big, other, b'@tail\nACGT'])
    assert parsed == len(big) + len(other) + 10
    got, (parsed, total) = plan([p1], batch)
    assert got == expected_plan([big]) and parsed == total == len(big)


def test_reader_stats_match_the_reference_on_gz(tmp_path):
    # parsed/total of the stored reference outcomes (float estimate for .gz, workhorse.c:883-884)
    for name in ('findseqs_gz', 'paired_gz', 'paired', 'multichunk_gz', 'ragged_two_files'):
        c = cases.by_name()[name]
        files = c.materialize(tmp_path)
        _, (parsed, total) = plan(files)
        st = expected()[name]['stats']
        assert (parsed, total) == (st['parsed'], st['total']), name


def test_reader_errors(tmp_path):
    got, err = plan([str(tmp_path / 'missing.fastq')])
    assert got is None and err[0] == _lib.ERR_IO and 'for getting filesize' in err[1]      # workhorse.c:661-668
    bad = str(tmp_path / 'bad.fastq.gz')
    open(bad, 'wb').write(b'this is not gzip')
    got, err = plan([bad])
    assert got is None and err[0] == _lib.ERR_IO and 'no valid gzip header' in err[1]      # workhorse.c:613-621
    # a full buffer without any record start (workhorse.c:921-929)
    junk = str(tmp_path / 'junk.fastq')
    open(junk, 'wb').write(b'A' * (3 << 20))
    got, err = plan([junk])
    assert got is None and err[0] == _lib.ERR_RUNTIME and 'could find beginning of record' in err[1]


def test_synthetic_generator_matches_numpy_statement():
    L = _lib.lib()
    g = synth.genome()
    raw = np.empty(4096, dtype=np.uint8)
    L.kvq_synth_genome_host(raw.ctypes.data, raw.nbytes, synth.SEED)
    planted = {p - 1 for p, _, _ in synth.RESISTANCE_SNPS}
    same = [i for i in range(4096) if i not in planted]
    assert (raw[same] == g[same]).all()
    for first, n, rl in ((0, 64, 150), (123456789, 33, 300), (7, 5, 25)):
        want = synth.reads(g, first, n, rl)
        got = np.empty(n * synth.record_bytes(rl), dtype=np.uint8)
        L.kvq_synth_reads_host(got.ctypes.data, first, n, rl, synth.SEED, g.ctypes.data, g.nbytes)
        assert (got == want).all()
    t = synth.table(g)
    assert len(t) == 132 and sum(map(len, t)) == 6318
    t2 = synth.table(g, 'MTBC+barcodes')
    assert len(t2) == 194 and sum(map(len, t2)) == 9480
    assert synth.revcomp(b'AACGTN') == b'NACGTT'                            # kvarq/genes.py:204,257-262


def test_reader_with_parallel_pread(tmp_path):
    """nthreads readers pread() slices of a plain file: same chunks, same bytes accounting"""
    big = cases.multichunk() * 4                     # > 4 MiB: the parallel path
    p = str(tmp_path / 'big.fastq')
    open(p, 'wb').write(big)
    saved = engine.get_config()
    try:
        engine.config(nthreads=5)
        got, (parsed, total) = plan([p], 3 << 20)
    finally:
        engine.config(**saved)
    assert got == expected_plan([big]) and parsed == total == len(big)


def bgzf(data, block=60000, level=6):
    """the bytes as a BGZF file (bgzip): gzip members of at most 64 KiB with the 'BC' extra subfield, then the empty EOF block"""
    import struct, zlib
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = co.compress(chunk) + co.flush()
        out.append(b'\x1f\x8b\x08\x04\0\0\0\0\x00\xff' + struct.pack('<H', 6) + b'BC' + struct.pack('<HH', 2, len(raw) + 25) +
                   raw + struct.pack('<II', zlib.crc32(chunk), len(chunk)))
    out.append(bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000'))
    return b''.join(out)


@pytest.mark.parametrize('nthreads', [1, 4])
def test_bgzf_blocks_are_inflated_in_parallel_to_the_same_stream(tmp_path, nthreads):
    """bgzip'ed FastQ (multi-member gzip, workhorse.c:842-866) gives the chunks of the plain text, also when
    ordinary gzip members follow the blocks (the serial reader takes over) and with trailing bytes"""
    L = _lib.lib()
    from kvarq_amd import engine
    engine.config(**dict(cases.PRODUCT, nthreads=nthreads))
    big = cases.multichunk()
    other = cases.ragged(34, 4000, cases.RAGGED_TARGETS)
    p1, p2, p3 = str(tmp_path / 'a.fastq.gz'), str(tmp_path / 'b.fastq.gz'), str(tmp_path / 'c.fastq.gz')
    z1 = bgzf(big)
    open(p1, 'wb').write(z1)
    z2 = bgzf(other[:len(other) // 2])[:-28] + gzip.compress(other[len(other) // 2:], mtime=0)      # blocks, then one plain member
    open(p2, 'wb').write(z2)
    open(p3, 'wb').write(bgzf(b'@tail\nACGT', block=7) + b'\0\0\0')                                   # tiny blocks, trailing bytes
    got, (parsed, total) = plan([p1, p2, p3])
    assert got == expected_plan([big, other, b'@tail\nACGT'])
    assert parsed == len(big) + len(other) + 10
    # chunks, parsed bytes and the final size estimate (float arithmetic, workhorse.c:883-884) == the serial reader's,
    # which the reference's own .gz outcomes pin (test_reader_stats_match_the_reference_on_gz)
    os.environ['KVQ_BGZF'] = '0'
    try:
        serial = plan([p1, p2, p3])
    finally:
        del os.environ['KVQ_BGZF']
    assert serial == (got, (parsed, total))
    got, (parsed, total) = plan([p1], 1 << 20)
    assert got == expected_plan([big]) and parsed == len(big)


@pytest.mark.parametrize('ahead_mb', [None, '4', '0'])
def test_the_second_gz_of_a_pair_is_inflated_ahead_to_the_same_stream(tmp_path, ahead_mb, monkeypatch):
    """plain .gz files are inflated by a reader thread each, the next file's running ahead of the stream
    (bounded by KVQ_GZ_AHEAD_MB): chunks, parsed bytes and the final size estimate == the one-thread reader's,
    which the reference's own .gz outcomes pin (test_reader_stats_match_the_reference_on_gz)"""
    a = cases.multichunk() * 3                      # 11.7 MB: several 4 MiB blocks queue up
    b = cases.ragged(34, 30000, cases.RAGGED_TARGETS)
    c = b'@tail\nACGT'
    p = [str(tmp_path / n) for n in ('r_1.fastq.gz', 'r_2.fastq.gz', 'r_3.fastq.gz', 'r_4.fastq')]
    open(p[0], 'wb').write(gzip.compress(a, 1, mtime=0))
    with open(p[1], 'wb') as f:                      # two members
        f.write(gzip.compress(b[:len(b) // 3], 1, mtime=0) + gzip.compress(b[len(b) // 3:], 1, mtime=0))
    open(p[2], 'wb').write(gzip.compress(c, mtime=0) + b'\0\0\0\0')
    open(p[3], 'wb').write(b)
    saved = engine.get_config()
    try:
        engine.config(nthreads=1)
        serial = plan(p, 3 << 20)
        if ahead_mb is not None:
            monkeypatch.setenv('KVQ_GZ_AHEAD_MB', ahead_mb)
        engine.config(nthreads=4)
        ahead = plan(p, 3 << 20)
        # a damaged second file: the error is raised when the stream gets there, with the stream position
        z = bytearray(open(p[1], 'rb').read())
        z[len(gzip.compress(b[:len(b) // 3], 1, mtime=0)) + 10] = 0x07           # the second member opens with block type 3: invalid
        open(p[1], 'wb').write(bytes(z))
        bad_ahead = plan(p[:2], 3 << 20)
        engine.config(nthreads=1)
        bad_serial = plan(p[:2], 3 << 20)
        nothing = str(tmp_path / 'x.fastq.gz')
        open(nothing, 'wb').write(b'no gzip')
        engine.config(nthreads=4)
        bad_header = plan([p[0], nothing])
    finally:
        engine.config(**saved)
    assert serial[0] == expected_plan([a, b, c, b]) and serial[1][0] == len(a) + 2 * len(b) + len(c)
    assert ahead == serial
    assert bad_serial[0] is None and bad_serial[1][0] == _lib.ERR_IO and 'error while inflating compressed data' in bad_serial[1][1]
    assert bad_serial[1][1].endswith('fpos=%d' % (len(a) + len(b) // 3))
    assert bad_ahead == bad_serial                                               # same status, same fpos
    assert bad_header[0] is None and 'no valid gzip header found at beginning of file' in bad_header[1][1]


def test_sigint_is_counted_by_a_c_handler_and_the_old_handler_comes_back():
    """workhorse.c:133-136, 1632: SIGINT only counts (stats()['sigints']); findseqs zeroes the count (1264-1265)"""
    import signal, time
    engine.install_sigint_counter()
    try:
        n0 = engine.stats()['sigints']
        os.kill(os.getpid(), signal.SIGINT)              # would raise KeyboardInterrupt under Python's own handler
        os.kill(os.getpid(), signal.SIGINT)
        time.sleep(0.05)
        assert engine.stats()['sigints'] == n0 + 2
    finally:
        engine.remove_sigint_counter()
    with pytest.raises(KeyboardInterrupt):               # Python's handler is back
        os.kill(os.getpid(), signal.SIGINT)
        time.sleep(1)


def test_bench_gpus_n_launches_n_ranks_before_any_hip_call():
    """`python bench.py --gpus N` (the driver's command shape) is a complete N-rank run: the launcher starts N fresh
    interpreters with the torch.distributed.run environment and has not loaded torch or libkvarq_hip itself -- the join
    of the reference's worker threads (workhorse.c:1375-1447) as a join of processes"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, KVQ_BENCH_CHILD_PROBE='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--total-reads', '3000'],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert 'launched 3 ranks' in p.stderr and 'before any HIP call' in p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line['rank'] == 0 and line['world'] == 3 and line['master'].startswith('127.0.0.1:')
    assert line['gpu_modules_loaded'] == []
    # a rank that fails takes the launcher's exit code with it
    env['KVQ_BENCH_CHILD_PROBE'] = 'fail'
    q = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env, capture_output=True, text=True, timeout=120)
    assert q.returncode == 3
