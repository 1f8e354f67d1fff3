"""
GPU parity tests (run with -m gpu on an MI355X): the HIP path, reached through
the C ABI exactly as a caller of ``kvarq.engine`` would reach it, against
 (1) the stored outcomes of the reference C engine (tests/golden/expected.json),
 (2) the oracle restatement run on the same inputs (also where no reference
     outcome exists: hits of length 0/1, coverage and mutation counters),
 (3) size-independent properties on device-resident synthetic data.
Bit-exact everywhere: hits, hit bytes, per-sequence counters, read-length
histogram, coverage and mutation counts.
"""
import os

import numpy as np
import pytest

import cases
from kvarq_amd import _lib, engine, scan, synth
from kvarq_amd.fastq import FastqFileFormatException
from oracle import oracle as O
from util import expected, check_against_expected

pytestmark = pytest.mark.gpu

# a tile that leaves records behind: only those go through the exhaustive kernels
REDO_PATH = dict(seeded=True, exhaustive=True, rescanned=False, tiles_rescanned=True)

CASES = cases.all_cases(big=True)


def run_engine(case, tmp_path):
    files = case.materialize(tmp_path)
    engine.config(**case.config)
    return files, engine.findseqs(files[0] if len(files) == 1 else files, case.seq_bytes())


@pytest.mark.parametrize('case', CASES, ids=lambda c: c.name)
def test_findseqs_matches_reference_and_oracle(case, tmp_path):
    exp = expected().get(case.name)
    if exp and 'error' in exp:
        kind, msg = exp['error']
        assert kind == 'format'
        with pytest.raises(FastqFileFormatException) as ei:
            run_engine(case, tmp_path)
        assert str(ei.value) == msg
        return
    files, r = run_engine(case, tmp_path)
    if exp:
        check_against_expected(case.name, r, exp)
    # and against the oracle on the same files: exact order, bytes and every stat
    o = O.findseqs(files[0] if len(files) == 1 else files, case.seq_bytes(), **dict(case.config, nthreads=2))
    assert tuple(r['hits']) == tuple(o['hits'])
    assert [bytes(h) for h in r['hitseqs']] == o['hitseqs']
    assert r['stats'] == o['stats']


def test_the_scan_kept_between_findseqs_calls_is_as_good_as_new(tmp_path, monkeypatch):
    """kvq_findseqs keeps its table and scan object for the next call with the same sequences and settings: files of
    other record sizes, a changed setting, other sequences and the same ones again must all give the oracle's answer
    (and the same with the keeping switched off)"""
    g = synth.genome()
    seqs = cases.mtbc_table()
    files = {}
    for name, (first, n, L) in dict(a=(0, 3000, 150), b=(5000, 2000, 300), c=(100, 4000, 75)).items():
        f = tmp_path / (name + '.fastq'); f.write_bytes(synth.reads(g, first, n, L).tobytes()); files[name] = str(f)
    plan = [('a', cases.PRODUCT, seqs), ('b', cases.PRODUCT, seqs), ('c', cases.PRODUCT, seqs), ('a', cases.PRODUCT, seqs),
            ('a', dict(cases.PRODUCT, maxerrors=1), seqs), ('b', dict(cases.PRODUCT, maxerrors=1), seqs[:40]), ('b', cases.PRODUCT, seqs)]
    want = [O.findseqs(files[f], q, **dict(cfg, nthreads=2)) for f, cfg, q in plan]
    for keep in ('1', '0'):
        monkeypatch.setenv('KVQ_KEEP_SCAN', keep)          # (read once per process: the second round repeats the first unless this process started with it)
        for (f, cfg, q), o in zip(plan, want):
            engine.config(**dict(cfg, nthreads=2))
            r = engine.findseqs(files[f], q)
            assert tuple(r['hits']) == tuple(o['hits']) and [bytes(h) for h in r['hitseqs']] == o['hitseqs'] and r['stats'] == o['stats'], (keep, f)
    _lib.lib().kvq_release_cached()                        # (drops the kept pair and the cached blocks; the next call starts from nothing)
    engine.config(**dict(cases.PRODUCT, nthreads=2))
    r = engine.findseqs(files['a'], seqs)
    assert tuple(r['hits']) == tuple(want[0]['hits']) and r['stats'] == want[0]['stats']


def test_str_and_bytes_interfaces(fastqs):
    engine.config(**dict(cases.DEFAULTS, maxerrors=0, minoverlap=1000, minreadlength=3, Amin='!'))
    f = fastqs + '/test_engine.fastq'
    a = engine.findseqs(f, cases.SEQS_FINDSEQS)
    b = engine.findseqs(f.encode(), [s.encode() for s in cases.SEQS_FINDSEQS])
    assert a['hits'] == b['hits'] and a['stats'] == b['stats']
    assert all(isinstance(h, str) for h in a['hitseqs']) and all(isinstance(h, bytes) for h in b['hitseqs'])
    assert [h.encode() for h in a['hitseqs']] == b['hitseqs']
    assert a['stats']['nseqhits'] == (19, 1, 0, 1, 1, 1, 1)               # test_engine.py:175
    assert isinstance(a['hits'][0], engine.Hit)
    s = engine.stats()                                                    # stats() after the scan == stats of the scan
    assert s['nseqhits'] == a['stats']['nseqhits'] and s['records_parsed'] == 14
    with pytest.raises(IOError):
        engine.findseqs(f + '.missing', ['ACGT'])
    with pytest.raises(TypeError):
        engine.findseqs(f, [1, 2])
    with pytest.raises(TypeError):
        engine.findseqs(12, ['ACGT'])


@pytest.mark.parametrize('name', ['ragged', 'ragged_tiny', 'spoligo_5k', 'quirk_e2', 'synth20k_mtbc'])
def test_coverage_and_mutation_counters_match_the_oracle_fold(name, tmp_path):
    case = cases.by_name()[name]
    files = case.materialize(tmp_path)
    data = b''.join(open(f, 'rb').read() for f in files)
    cfg = {k: v for k, v in case.config.items()}
    o = O.scan_memory(data, case.seq_bytes(), fold=True, **dict(cfg, nthreads=2)) if len(files) == 1 else None
    if o is None:
        pytest.skip('single-stream cases only')
    t = scan.Table(case.seq_bytes(), **cfg)
    s = scan.Scanner(t)
    s.scan_host(data)
    r = s.finish()
    assert tuple(r['hits']) == tuple(o['hits'])
    assert r['hitseqs'] == o['hitseqs']
    assert r['stats']['nseqhits'] == o['stats']['nseqhits'] and r['stats']['readlengths'] == o['stats']['readlengths']
    assert r['coverage'].tolist() == o['coverage']
    assert r['mutations'].tolist() == o['mutations']
    assert int(r['counters'][_lib.CTR_HITS]) == len(o['hits'])
    s.close()
    t.close()


def test_device_resident_synthetic_scan():
    """reads generated on the device == numpy statement; resident scan == oracle; rescans are idempotent"""
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    n, L = 30000, 150
    rb = synth.record_bytes(L)
    dg = scan.DeviceBuffer(g.nbytes)
    dg.upload(g)
    dd = scan.DeviceBuffer(n * rb)
    assert _lib.lib().kvq_synth_reads_device(dd.ptr, 5000, n, L, synth.SEED, dg.ptr, g.nbytes) == 0
    host = synth.reads(g, 5000, n, L)
    assert (dd.download() == host).all()
    cfg = dict(cases.PRODUCT)
    o = O.scan_memory(host, seqs, fpos_base=5000 * rb, fold=True, **dict(cfg, nthreads=8))
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    co = scan.chunk_offsets(host)
    for rep in range(2):
        s.scan_device(dd.ptr, host.nbytes, co, fpos_base=5000 * rb)
        r = s.finish()
        assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
        assert r['coverage'].tolist() == o['coverage'] and r['mutations'].tolist() == o['mutations']
        assert r['stats']['records_parsed'] == n and r['stats']['readlengths'] == o['stats']['readlengths']
        s.reset()
    # two half batches == one batch (file_pos stays global)
    cut = int(co[len(co) // 2])
    s.scan_device(dd.ptr, cut, co[:len(co) // 2 + 1], fpos_base=5000 * rb)
    assert cut % 16 == 0 or True
    half2 = host[cut:]
    d2 = scan.DeviceBuffer(half2.nbytes)
    d2.upload(half2)
    s.scan_device(d2.ptr, half2.nbytes, co[len(co) // 2:] - cut, fpos_base=5000 * rb + cut)
    r = s.finish()
    assert tuple(r['hits']) == tuple(o['hits']) and r['coverage'].tolist() == o['coverage']
    s.close(); t.close(); dd.free(); dg.free(); d2.free()


def test_a_batch_beyond_2_gib_equals_the_same_text_in_two_batches():
    """offsets inside a batch are 32-bit: 7.2 M records = 2.34 GB in one call == the same records in two calls"""
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    n, L = 7_200_000, 150
    rb = synth.record_bytes(L)
    dg = scan.DeviceBuffer(g.nbytes)
    dg.upload(g)
    dd = scan.DeviceBuffer(n * rb)
    assert n * rb > (1 << 31)
    assert _lib.lib().kvq_synth_reads_device(dd.ptr, 0, n, L, synth.SEED, dg.ptr, g.nbytes) == 0
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    co = bench.analytic_chunk_offsets(n, rb, L)
    t = scan.Table(seqs, **cases.PRODUCT)
    s = scan.Scanner(t)
    s.scan_device(dd.ptr, n * rb, co)
    one = s.finish()
    s.reset()
    h = len(co) // 2
    cut = int(co[h])
    assert cut % 16 == 0 or True
    base2 = cut & ~15
    s.scan_device(dd.ptr, cut, co[:h + 1])
    s.scan_device(dd.ptr + base2, n * rb - base2, co[h:] - base2, fpos_base=base2)
    two = s.finish()
    assert one['stats']['records_parsed'] == two['stats']['records_parsed'] == n
    assert one['n_hits'] == two['n_hits'] > 20000 and one['hits'] == two['hits'] and one['hitseqs'] == two['hitseqs']
    assert (one['counters'] == two['counters']).all()
    assert max(h.file_pos for h in one['hits']) > (1 << 31)
    assert sum(one['stats']['readlengths']) == n
    # ... and == the exhaustive kernels on the same 2.3 GB (every workgroup walks ~110 tiles drawn from the
    # tile counter and flushes its read-length histogram on the way; nothing of that exists on the other path)
    s.reset()
    s.force_exhaustive(True)
    s.scan_device(dd.ptr, n * rb, co)
    three = s.finish()
    assert three['path'] == dict(seeded=False, exhaustive=True, rescanned=False, tiles_rescanned=False)
    assert one['hits'] == three['hits'] and one['hitseqs'] == three['hitseqs'] and (one['counters'] == three['counters']).all()
    s.close(); t.close(); dd.free(); dg.free()


def test_bgzf_file_equals_the_oracle_on_the_same_file(tmp_path):
    """a bgzip'ed FastQ (inflated block-parallel on the host) against the oracle's serial zlib reader: hits, bytes, all stats"""
    from test_host_logic import bgzf
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    data = synth.reads(g, 777, 40000, 150).tobytes()
    p = str(tmp_path / 'reads.fastq.gz')
    open(p, 'wb').write(bgzf(data, level=1))
    for nt in (1, 8):
        cfg = dict(cases.PRODUCT, nthreads=nt)
        engine.config(**cfg)
        r = engine.findseqs(p, seqs)
        o = O.findseqs(p, seqs, **cfg)
        assert tuple(r['hits']) == tuple(o['hits']) and len(r['hits']) > 50
        assert [bytes(h) for h in r['hitseqs']] == o['hitseqs']
        assert r['stats'] == o['stats']


def test_hit_arena_overflow_is_transparent():
    """more hits than the initial arena holds: rescan with a larger one, same result"""
    read = 'ACG' * 60
    data = cases.rec('x', read, 'I' * len(read)) * 40000        # 60 'ACG' hits per record -> 2.4M hits
    seqs = [b'ACG']
    cfg = dict(cases.DEFAULTS, minreadlength=10)
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(len(data))
    arr = np.frombuffer(data, dtype=np.uint8)
    d.upload(arr)
    s.scan_device(d.ptr, arr.nbytes, scan.chunk_offsets(arr))
    r = s.finish(hits=False)
    assert r['n_hits'] == 40000 * 60
    assert int(r['counters'][t.off_nseqhits]) == 40000 * 60
    assert r['coverage'].tolist() == [40000 * 60] * 3
    s.close(); d.free()
    # the same through host batches: the library cannot replay what it no longer holds (KVQ_ERR_RESCAN,
    # include/kvarq_hip.h); Scanner.finish feeds its batches again to the enlarged arena
    s = scan.Scanner(t)
    half = (len(data) // 2 // len(cases.rec('x', read, 'I' * len(read)))) * len(cases.rec('x', read, 'I' * len(read)))
    s.scan_host(arr[:half])
    s.scan_host(arr[half:], fpos_base=half)
    r = s.finish(hits=False)
    assert r['n_hits'] == 40000 * 60 and int(r['counters'][t.off_nseqhits]) == 40000 * 60
    assert r['coverage'].tolist() == [40000 * 60] * 3
    fp = scan._view(_lib.lib().kvq_scan_hit_file_pos(s.h), r['n_hits'], __import__('ctypes').c_int64, np.int64)
    assert int(fp[0]) == 3 and int(fp[-1]) == len(data) - len(cases.rec('x', read, 'I' * len(read))) + 3 and bool((np.diff(fp) >= 0).all())      # ('@x\\n': the bases begin three bytes into a record)
    # and a C caller that does not replay is told so
    s2 = scan.Scanner(t)
    L = _lib.lib()
    co = scan.chunk_offsets(arr)
    assert L.kvq_scan_host(s2.h, arr.ctypes.data, arr.nbytes, co.ctypes.data_as(__import__('ctypes').POINTER(__import__('ctypes').c_int64)), len(co) - 1, 0) == 0
    rc = L.kvq_scan_finish(s2.h)
    assert rc != 0 and _lib.last_error()[0] == _lib.ERR_RESCAN
    s2.close(); s.close(); t.close()


def test_many_hits_per_read_come_back_in_reference_order():
    """dozens of hits share one file position (the bucket ordering gives up and the comparison sort takes
    over): order, hit bytes and counters still equal the oracle's"""
    read = 'ACG' * 60
    data = np.frombuffer(cases.rec('x', read, 'I' * len(read)) * 300 + cases.rec('y', 'TTTTACGTTTTTTTTT', 'I' * 16) * 50, dtype=np.uint8)
    seqs = [b'ACG', b'CGA', b'GAC', b'TTACG']
    cfg = dict(cases.DEFAULTS, minreadlength=10)
    o = O.scan_memory(data, seqs, fold=True, **dict(cfg, nthreads=4))
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(data.nbytes)
    d.upload(data)
    s.scan_device(d.ptr, data.nbytes, scan.chunk_offsets(data))
    r = s.finish()
    assert r['n_hits'] == len(o['hits']) > 300 * 170
    assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
    assert r['coverage'].tolist() == o['coverage'] and r['mutations'].tolist() == o['mutations']
    s.close(); t.close(); d.free()


@pytest.mark.parametrize('tile', [None, '8000', '8080', '24000'])
def test_well_formed_files_never_fall_back_to_the_exhaustive_kernels(tile, monkeypatch):
    """ragged records over several chunks, at several tile sizes: every tile's speculated record
    split is accepted (a rejected one costs a rescan of the whole batch with the exhaustive
    kernels -- the chunk's last tile, which often owns nothing but the final newline, used to be one)"""
    import os
    import random
    if tile is not None:
        monkeypatch.setenv('KVQ_TILE', tile)
    rng = random.Random(77)
    genome = cases.randseq(rng, 5000)
    recs = []
    for i in range(26000):
        L = rng.choice([76, 100, 125, 150, 150, 151, rng.randint(40, 250)])
        a = rng.randrange(0, len(genome) - L)
        recs.append(cases.rec('M01:%d:%d:%d' % (i, rng.randrange(30000), rng.randrange(30000)), genome[a:a + L], 'I' * L))
    text = b''.join(recs)
    assert len(text) > 5 * (1 << 20)
    data = np.frombuffer(text, dtype=np.uint8)
    seqs = [genome[100:151].encode(), genome[2000:2051].encode(), genome[3000:3025].encode()]
    t = scan.Table(seqs, **cases.PRODUCT)
    res = []
    for force in (False, True):
        s = scan.Scanner(t)
        s.force_exhaustive(force)
        s.scan_host(data)
        res.append(s.finish())
        s.close()
    a, b = res
    assert a['path'] == dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False), a['path']
    assert a['n_hits'] == b['n_hits'] > 100 and tuple(a['hits']) == tuple(b['hits']) and a['hitseqs'] == b['hitseqs']
    assert a['counters'].tolist() == b['counters'].tolist()
    t.close()


@pytest.mark.parametrize('k', [4, 7, 12, 20])
def test_a_few_dozen_hits_per_read_are_ranked_by_the_wave(k):
    """9 to 64 hits on one file position: the bucket ordering ranks them with one lane per hit
    (kernels_results.hip): 10, 19, 34 and 58 hits per read"""
    read = 'ACG' * k + 'T'
    data = np.frombuffer(b''.join(cases.rec('r%d' % i, read, 'I' * len(read)) for i in range(400)), dtype=np.uint8)
    seqs = [b'ACG', b'CGA', b'GAC']
    cfg = dict(cases.DEFAULTS, minreadlength=10)
    o = O.scan_memory(data, seqs, fold=True, **dict(cfg, nthreads=4))
    per_read = len(o['hits']) // 400
    assert 8 < per_read <= 64
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(data.nbytes)
    d.upload(data)
    s.scan_device(d.ptr, data.nbytes, scan.chunk_offsets(data))
    r = s.finish()
    assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
    assert r['coverage'].tolist() == o['coverage']
    s.close(); t.close(); d.free()


def _wait_until_scanning(th, total, seconds=60):
    """blocks until the scan (of ``total`` stream bytes) running in ``th`` has published the counters of a batch"""
    import time
    t0 = time.time()
    while th.is_alive() and time.time() - t0 < seconds:
        st = engine.stats()
        if st['total'] == total and st['records_parsed'] > 0:      # (before it opens its files the last scan's state shows)
            return True
        time.sleep(0.0005)
    return False


def test_concurrent_findseqs_is_refused_and_stop_works(tmp_path, monkeypatch):
    """a second findseqs while one runs raises (workhorse.c:1258-1262); stop() ends the running one at the next
    batch with the hits found so far and no error (workhorse.c:1469-1479) -- BOTH must be seen"""
    import threading
    monkeypatch.setenv('KVQ_BATCH_MB', '2')                   # ~80 batches out of 156 MB: the scan cannot be over after two
    p = tmp_path / 'big.fastq'
    p.write_bytes(cases.multichunk() * 40)
    total_records, total_bytes = 9000 * 40, 40 * len(cases.multichunk())
    engine.config(**dict(cases.PRODUCT, nthreads=1))
    out = {}

    def run():
        try:
            out['r'] = engine.findseqs(str(p), cases.MULTI_SEQS)
        except Exception as e:                                 # (would mean the main thread's call won the race)
            out['e'] = e
    th = threading.Thread(target=run)
    th.start()
    seen = _wait_until_scanning(th, total_bytes)
    assert seen, 'the scan never published a batch: %r %r' % (out.get('e'), engine.stats())
    with pytest.raises(RuntimeError, match='already running'):
        engine.findseqs(str(p), cases.MULTI_SEQS)
    engine.stop()
    th.join()
    assert 'e' not in out, out.get('e')
    r = out['r']
    st = r['stats']
    assert 0 < st['records_parsed'] < total_records          # partial, and not empty
    assert 0 < st['parsed'] < st['total'] == total_bytes
    assert sum(st['readlengths']) == st['records_parsed']
    assert len(r['hits']) == len(r['hitseqs']) == sum(st['nseqhits']) > 0
    # what it did return is a prefix of the full scan: the hits of the records before the stop, in file order
    full = engine.findseqs(str(p), cases.MULTI_SEQS)
    assert full['stats']['records_parsed'] == total_records
    assert tuple(full['hits'][:len(r['hits'])]) == tuple(r['hits'])
    assert full['hitseqs'][:len(r['hits'])] == r['hitseqs']


def test_stats_polled_from_another_thread_grow_monotonically(tmp_path, monkeypatch):
    """engine.stats() during a scan of many batches (workhorse.c:1205-1244 reads the worker's counters while it
    runs): every field only ever grows, it ends on the scan's own stats, and more than two distinct states are seen"""
    import threading, time
    monkeypatch.setenv('KVQ_BATCH_MB', '2')
    p = tmp_path / 'big.fastq'
    p.write_bytes(cases.multichunk() * 30)
    engine.config(**dict(cases.PRODUCT, nthreads=1))
    out = {}
    th = threading.Thread(target=lambda: out.setdefault('r', engine.findseqs(str(p), cases.MULTI_SEQS)))
    seen = []
    th.start()
    while th.is_alive():
        seen.append(engine.stats())
        time.sleep(0.0005)
    th.join()
    seen.append(engine.stats())
    final = out['r']['stats']
    assert seen[-1] == final
    assert final['records_parsed'] == 9000 * 30 and final['progress'] == 1.0
    started = [s for s in seen if s['total'] == final['total']]       # (polls before the scan opened its files see the last scan's state)
    for a, b in zip(started, started[1:]):
        assert a['records_parsed'] <= b['records_parsed']
        assert a['parsed'] <= b['parsed'] and a['progress'] <= b['progress']
        assert all(x <= y for x, y in zip(a['nseqhits'], b['nseqhits']))
        assert all(x <= y for x, y in zip(a['nseqbasehits'], b['nseqbasehits']))
        assert sum(a['readlengths']) == a['records_parsed']              # one consistent snapshot, not a torn one
    assert len(set(s['records_parsed'] for s in started)) > 3


def test_seeded_and_exhaustive_kernels_agree_and_paths_are_as_expected(tmp_path):
    """the seed-filter kernel against the exhaustive kernels on the same device data,
    and which of them served each case"""
    want_path = {
        'synth20k_mtbc': dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False),
        'spoligo_5k': dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False),
        'multichunk': dict(seeded=True, exhaustive=True, rescanned=False, tiles_rescanned=False),      # N-holding sequence -> exhaustive
        'quirk_e2': dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False),
        'long_reads': dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False),     # 11 kB in all: one tile holds it
        'ragged_crlf': dict(seeded=False, exhaustive=True, rescanned=False, tiles_rescanned=False),    # e=3, minoverlap 12: not seedable
        'findseqs': dict(seeded=False, exhaustive=True, rescanned=False, tiles_rescanned=False),
        # seeds shorter than 8 where (maxerrors + 1) * 8 does not fit the shortest accepted overlap (kvq_seed_k): the reference's own
        # maxerrors sweep at minoverlap 25 (test_engine.py:208-224: e = 3 -> K = 6) and its analyser settings (test_analyser.py:55-58:
        # minoverlap 10, maxerrors 1 -> K = 5) stay on the seed filter
        'maxerror3': dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False),
        'maxerror2': dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False),
        'spoligo_analyser': dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False),
    }
    want_k = {'maxerror3': 6, 'maxerror2': 8, 'spoligo_analyser': 5, 'synth20k_mtbc': 8}
    # a 9 kB record that starts 36 kB into the text outgrows the look-ahead of the tile that owns it
    # (a tile and its look-ahead span 40.8 kB): that tile's records are scanned again by the exhaustive kernels
    import os
    if not os.environ.get('KVQ_TILE'):   # (the bit-plane kernel, or another tile size, cuts the tiles elsewhere)
        want_path['long_reads_straddle'] = REDO_PATH   # (the one tile's records only)
    for name, wp in want_path.items():
        case = cases.by_name()[name.replace('_straddle', '')]
        files = case.materialize(tmp_path)
        text = b''.join(open(f, 'rb').read() for f in files[:1])
        if name.endswith('_straddle'):
            import random
            rng = random.Random(5)
            filler = b''
            while len(filler) < 36000 - 400:
                b = cases.randseq(rng, 150)
                filler += cases.rec('f%d' % len(filler), b, 'I' * 150)
            big = cases.randseq(rng, 4400)
            text = filler + cases.rec('big', big, 'I' * 4400) + text
        data = np.frombuffer(text, dtype=np.uint8)
        t = scan.Table(case.seq_bytes(), **case.config)
        if name in want_k and not os.environ.get('KVQ_K'):
            assert t.seed_k == want_k[name], (name, t.seed_k)
        res = []
        for force in (False, True):
            s = scan.Scanner(t)
            s.force_exhaustive(force)
            s.scan_host(data)
            res.append(s.finish())
            s.close()
        a, b = res
        assert a['path'] == wp, (name, a['path'])
        assert b['path']['seeded'] is False
        assert a['hits'] == b['hits'] and a['hitseqs'] == b['hitseqs'], name
        assert (a['counters'] == b['counters']).all(), name
        t.close()


def test_an_empty_or_refused_batch_leaves_no_hole():
    """scan_device(A), scan_device(nothing), a refused batch, scan_device(B) == one scan of A + B: a batch that
    scans nothing must not leave the range of hits of the batches behind it open"""
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    host = synth.reads(g, 0, 40000, 150)
    co = scan.chunk_offsets(host)
    cut = int(co[len(co) // 2])
    t = scan.Table(seqs, **cases.PRODUCT)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(host.nbytes)
    d.upload(host)
    s.scan_device(d.ptr, host.nbytes, co)
    whole = s.finish()
    s.reset()
    d2 = scan.DeviceBuffer(host.nbytes - cut)
    d2.upload(host[cut:])
    s.scan_device(d.ptr, cut, co[:len(co) // 2 + 1])
    s.scan_device(d.ptr, 0, np.zeros(1, dtype=np.int64))                         # nothing
    with pytest.raises(RuntimeError):
        s.scan_device(d.ptr, cut, np.array([0, cut + 5], dtype=np.int64))        # bad chunk offsets: refused
    s.scan_device(d2.ptr, host.nbytes - cut, co[len(co) // 2:] - cut, fpos_base=cut)
    parts = s.finish()
    assert whole['n_hits'] > 50 and tuple(parts['hits']) == tuple(whole['hits']) and parts['hitseqs'] == whole['hitseqs']
    assert parts['counters'].tolist() == whole['counters'].tolist()
    s.close(); t.close(); d.free(); d2.free()


def test_a_batch_of_empty_chunks_behind_a_batch_with_skipped_tiles():
    """A batch whose chunks are all empty scans no tile; the redo chain behind it must not run on the counts the batch in
    front of it left (a tile of that batch skipped its 5 kB record: one record on the redo's list)."""
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    plain = synth.reads(g, 0, 2000, 150)
    big = bytes(g[1000:6000])
    rec = b'@big 1:N:0\n' + big + b'\n+\n' + b'I' * len(big) + b'\n'
    rb = synth.record_bytes(150)
    text = np.frombuffer(plain[:108 * rb].tobytes() + rec + plain[108 * rb:].tobytes(), dtype=np.uint8)
    want = O.scan_memory(text, seqs, nthreads=2, **cases.PRODUCT)
    t = scan.Table(seqs, **cases.PRODUCT)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(text.nbytes); d.upload(text)
    s.scan_device(d.ptr, text.nbytes, scan.chunk_offsets(text))
    s.scan_device(d.ptr, 0, np.zeros(2, dtype=np.int64), fpos_base=text.nbytes)       # one empty chunk
    s.scan_device(d.ptr, 0, np.zeros(3, dtype=np.int64), fpos_base=text.nbytes)       # two of them
    r = s.finish()
    assert r['path']['tiles_rescanned'] and r['stats']['records_parsed'] == 2001
    assert tuple(r['hits']) == tuple(want['hits']) and r['stats']['nseqhits'] == want['stats']['nseqhits']
    assert r['stats']['readlengths'] == want['stats']['readlengths']
    s.close(); t.close(); d.free()


def test_native_rccl_communicator_of_one_rank():
    """the library's own RCCL entry points (include/kvarq_hip.h, "several GPUs") on the one GPU there is: a
    communicator of one rank; finish sums the counters over it (all-reduce, maximum for the longest read) and
    gather_hits collects the hit arrays (all-gather of the counts, a broadcast per array) -- both must leave a
    single rank's result as it is.  (More ranks need more GPUs: the two-rank logic runs over gloo on CPUs.)"""
    from kvarq_amd import dist as kdist
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    host = synth.reads(g, 0, 20000, 150)
    t = scan.Table(seqs, **cases.PRODUCT)
    s = scan.Scanner(t)
    s.scan_host(host)
    alone = s.finish()
    comm = kdist.NativeComm(1, 0, lambda uid: uid)
    s.reset()
    s.set_comm(comm)
    s.scan_host(host)
    r = s.finish()
    assert r['n_hits'] == alone['n_hits'] > 30 and tuple(r['hits']) == tuple(alone['hits'])
    assert r['counters'].tolist() == alone['counters'].tolist()
    gathered = s.gather_hits()
    assert tuple(gathered['hits']) == tuple(alone['hits']) and gathered['hitseqs'] == alone['hitseqs']
    s.set_comm(None)
    s.close(); t.close(); comm.close()

def _ranks_in_threads(world, work):
    """`work(rank, comm)` on `world` threads of this process, each with a loopback communicator of its own
    (kvq_comm_create_local): the library's join code with several ranks on the one GPU there is"""
    import threading
    from kvarq_amd import dist as kdist
    key = int.from_bytes(os.urandom(7), 'little')
    out, err = [None] * world, [None] * world
    def run(r):
        comm = None
        try:
            comm = kdist.LocalComm(world, r, key)
            out[r] = work(r, comm)
        except BaseException as e:                                  # noqa: a failure of one rank must not leave the others waiting unnoticed
            err[r] = e
        finally:
            if comm is not None:
                comm.close()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(300)
    assert not any(x.is_alive() for x in th), 'a rank is stuck in a collective'
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize('world', [2, 3])
def test_native_join_of_several_ranks_equals_one_scan(world):
    """kvq_scan_finish + kvq_scan_gather_hits with MORE THAN ONE rank (workhorse.c:1398-1447): the ranks are threads
    with a scan each, their communicator the library's loopback -- counters summed, longest read the maximum,
    hits of all ranks in stream order on every rank, equal to the oracle's single scan; finishing twice gives
    the same sums (the ranks' own counters are not overwritten)"""
    from kvarq_amd import dist as kdist
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    nreads, rb = 9000, synth.record_bytes(150)
    host = synth.reads(g, 0, nreads, 150)
    whole = O.scan_memory(host, seqs, fold=True, nthreads=4, **cases.PRODUCT)
    t = scan.Table(seqs, **cases.PRODUCT)
    def work(rank, comm):
        lo, hi = kdist.shard(nreads, rank, world)
        if world == 3 and rank == 1:
            hi = lo                                                   # a rank without a single read
        if world == 3 and rank == 2:
            lo = kdist.shard(nreads, 1, world)[0]
        s = scan.Scanner(t)
        s.set_comm(comm)
        s.scan_host(host[lo * rb:hi * rb], fpos_base=lo * rb)
        r = s.finish(hits=False)
        again = s.finish(hits=False)                                  # (collective too: every rank does it)
        assert again['counters'].tolist() == r['counters'].tolist()
        got = s.gather_hits()
        got2 = s.gather_hits()                                        # a second call finds the gathered arrays in place
        assert got2['n_hits'] == got['n_hits']
        res = dict(counters=r['counters'].copy(), hits=[tuple(h) for h in got['hits']], hitseqs=[bytes(x) for x in got['hitseqs']], stats=r['stats'],
                   coverage=r['coverage'].tolist(), mutations=r['mutations'].tolist())
        s.set_comm(None); s.close()
        return res
    outs = _ranks_in_threads(world, work)
    for o in outs:
        assert o['hits'] == [tuple(h) for h in whole['hits']]
        assert o['hitseqs'] == [x if isinstance(x, bytes) else x.encode('latin-1') for x in whole['hitseqs']]
        assert o['stats']['records_parsed'] == nreads and o['stats']['nseqhits'] == whole['stats']['nseqhits']
        assert o['stats']['nseqbasehits'] == whole['stats']['nseqbasehits'] and o['stats']['readlengths'] == whole['stats']['readlengths']
        assert o['coverage'] == whole['coverage'] and o['mutations'] == whole['mutations']
    t.close()


def test_arena_overflow_on_one_rank_takes_every_rank_round_again():
    """rank 1's host batches overflow its hit arena (KVQ_ERR_RESCAN), rank 0's do not: BOTH ranks are told to go round
    again, no partial sum is taken, their collectives stay in step, and the second round gives the right totals"""
    read = 'ACG' * 60
    rec_hits = cases.rec('x', read, 'I' * len(read))
    rec_none = cases.rec('x', 'T' * len(read), 'I' * len(read))
    shares = [rec_none * 20000 + rec_hits * 10, rec_hits * 40000]          # 600 hits / 2.4 M hits (the arena starts at 1 M)
    t = scan.Table([b'ACG'], **dict(cases.DEFAULTS, minreadlength=10))
    def work(rank, comm):
        s = scan.Scanner(t, retain_limit=0)                                # (the caller replays: it sees the KVQ_ERR_RESCAN of the other rank)
        s.set_comm(comm)
        rounds = 0
        while True:
            rounds += 1
            s.scan_host(np.frombuffer(shares[rank], dtype=np.uint8), fpos_base=0 if rank == 0 else len(shares[0]))
            try:
                r = s.finish(hits=False)
                break
            except scan.RescanRequired:
                s.reset()
        n = int(r['counters'][t.off_nseqhits])
        own = int(r['n_hits'])
        s.set_comm(None); s.close()
        return rounds, n, own, int(r['counters'][_lib.CTR_RECORDS])
    outs = _ranks_in_threads(2, work)
    assert [o[0] for o in outs] == [2, 2]
    assert [o[1] for o in outs] == [600 + 40000 * 60] * 2 and [o[2] for o in outs] == [600, 40000 * 60]
    assert [o[3] for o in outs] == [60010] * 2
    t.close()


def test_replay_of_kept_host_batches_neither_copies_them_again_nor_forgets_them():
    """Scanner.finish feeds its kept copies again when the hit arena overflows.  The copies are kept once: with a retain limit
    between one and two times the data the replay must not count them a second time (and drop them, so that a further
    overflow could not be served), and reset() starts the count afresh however often a Scanner is reused."""
    read = 'ACG' * 60
    data = np.frombuffer(cases.rec('x', read, 'I' * len(read)) * 40000, dtype=np.uint8)          # 2.4 M hits: the arena starts at 1 M
    t = scan.Table([b'ACG'], **dict(cases.DEFAULTS, minreadlength=10))
    s = scan.Scanner(t, retain_limit=data.nbytes * 3 // 2)
    for round_ in range(3):                                                    # (the arena stays enlarged: only the first round overflows)
        s.scan_host(data)
        r = s.finish(hits=False)
        assert int(r['counters'][t.off_nseqhits]) == 40000 * 60 and int(r['counters'][_lib.CTR_RECORDS]) == 40000
        assert s._host_batches is not None and len(s._host_batches) == 1 and s._retained == data.nbytes
        s.reset()
        assert s._host_batches == [] and s._retained == 0
    # a device batch goes round again too (the caller's memory is named again, not copied)
    d = scan.DeviceBuffer(data.nbytes); d.upload(data)
    t2 = scan.Table([b'ACG', b'CGA'], **dict(cases.DEFAULTS, minreadlength=10))      # (a fresh table and scan: a fresh arena)
    s2 = scan.Scanner(t2)
    s2.scan_host(data[:len(data) // 2], fpos_base=0)
    s2.scan_device(d.ptr + 0, data.nbytes, scan.chunk_offsets(data), fpos_base=data.nbytes)
    r2 = s2.finish(hits=False)
    assert int(r2['counters'][_lib.CTR_RECORDS]) == 60000 and int(r2['counters'][t2.off_nseqhits]) == 60000 * 60
    s.close(); s2.close(); t.close(); t2.close(); d.free()


@pytest.mark.parametrize('cap', [0, 16, 100, 5000])
def test_a_full_survivors_list_costs_speed_never_results(cap, monkeypatch):
    """Work items that pass the scan kernel's 16-base test go to a list that kvq_verify_survivors empties behind the kernel, a wave taking
    its slots sixteen at a time; what a full list has no room for is verified in place.  With the list cut down to nothing, to one chunk,
    to a few, to less than the input needs (KVQ_SURV_CAP: slots), the hits, their order and every counter stay the oracle's."""
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    # reads sampled from the templates' own loci: every read is a true hit on several diagonals (a few thousand survivors)
    host = synth.reads(g, 0, 30000, 150)
    want = O.scan_memory(host, seqs, fold=True, nthreads=4, **cases.PRODUCT)
    monkeypatch.setenv('KVQ_SURV_CAP', str(cap))
    t = scan.Table(seqs, **cases.PRODUCT)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(host.nbytes); d.upload(host)
    for rep in range(2):                                               # (the list's counter starts afresh with every launch)
        s.reset()
        s.scan_device(d.ptr, host.nbytes, scan.chunk_offsets(host))
        r = s.finish()
        assert len(want['hits']) > 50 and tuple(r['hits']) == tuple(want['hits']) and r['hitseqs'] == want['hitseqs']
        assert r['stats']['nseqhits'] == want['stats']['nseqhits'] and r['stats']['nseqbasehits'] == want['stats']['nseqbasehits']
        assert r['coverage'].tolist() == want['coverage'] and r['mutations'].tolist() == want['mutations']
    s.close(); t.close(); d.free()


def test_the_tail_of_a_scan_can_be_enqueued_ahead_of_its_finish():
    """kvq_scan_finish_begin puts the ordering of the hits and the copies to the host on the stream behind the scan's kernels and returns;
    kvq_scan_finish then only waits.  Results are those of a plain finish -- also when another batch is fed behind the early tail (finish
    enqueues the tail again), when it is called twice, on a scan without batches, and with three scanners taking turns as bench.py's do."""
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    host = synth.reads(g, 0, 60000, 150)
    want = O.scan_memory(host, seqs, fold=True, nthreads=4, **cases.PRODUCT)
    assert len(want['hits']) > 50
    co = scan.chunk_offsets(host)
    half = int(co[len(co) // 2])
    co_a = co[:len(co) // 2 + 1]; co_b = co[len(co) // 2:] - half
    t = scan.Table(seqs, **cases.PRODUCT)
    d = scan.DeviceBuffer(host.nbytes); d.upload(host)

    def same(r):
        assert tuple(r['hits']) == tuple(want['hits']) and r['hitseqs'] == want['hitseqs']
        assert r['stats']['nseqhits'] == want['stats']['nseqhits'] and r['stats']['records_parsed'] == want['stats']['records_parsed']
        assert r['stats']['readlengths'] == want['stats']['readlengths']
        assert r['coverage'].tolist() == want['coverage'] and r['mutations'].tolist() == want['mutations']

    s = scan.Scanner(t)
    s.finish_begin()                                                   # (nothing fed yet: harmless)
    s.scan_device(d.ptr, host.nbytes, co)
    s.finish_begin(); s.finish_begin()
    same(s.finish())
    s.reset()
    d2 = scan.DeviceBuffer(host.nbytes - half); d2.upload(host[half:])  # (a batch starts on a 16-byte boundary)
    s.scan_device(d.ptr, half, co_a)
    s.finish_begin()                                                   # ... and then another batch after all
    s.scan_device(d2.ptr, host.nbytes - half, co_b, fpos_base=half)
    same(s.finish())
    s.reset()
    s.scan_host(host[:half], fpos_base=0)                              # (host batches: the one in flight is settled first)
    s.finish_begin()
    s.scan_host(host[half:], fpos_base=half)
    s.finish_begin()
    same(s.finish())
    ring = [s, scan.Scanner(t), scan.Scanner(t)]
    flying = []
    for i in range(7):
        sc = ring[i % 3]; sc.reset(); sc.scan_device(d.ptr, host.nbytes, co); sc.finish_begin(); flying.append(sc)
        if len(flying) == 3:
            same(flying.pop(0).finish())
    while flying:
        same(flying.pop(0).finish())
    for sc in ring:
        sc.close()
    t.close(); d.free(); d2.free()


def test_one_long_record_costs_its_tile_not_the_batch():
    """300 k ordinary reads with ONE 5 kB record in their middle: the record outgrows the look-ahead of the tile
    that owns it; only that tile's records go through the exhaustive kernels (path: tiles_rescanned, not
    rescanned), the result equals the oracle's, and the scan takes about as long as without the record"""
    import time
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    n, L = 300000, 150
    rb = synth.record_bytes(L)
    plain = synth.reads(g, 0, n, L)
    big = bytes(g[1000:6000])                                    # 5000 bases of the genome
    rec = b'@big 1:N:0\n' + big + b'\n+\n' + b'I' * len(big) + b'\n'
    # (108 records = 35 100 bytes into a chunk: inside its first tile of 39 760 bytes + 1 040 bytes of look-ahead,
    # the record's last newline 10 kB further on)
    co_plain = scan.chunk_offsets(plain)
    cut = int(co_plain[len(co_plain) // 2]) + 108 * rb
    text = np.frombuffer(plain[:cut].tobytes() + rec + plain[cut:].tobytes(), dtype=np.uint8)
    cfg = dict(cases.PRODUCT)
    o = O.scan_memory(text, seqs, fold=True, **dict(cfg, nthreads=16))
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    times = {}
    for name, data in (('plain', plain), ('long', text)):
        d = scan.DeviceBuffer(data.nbytes)
        d.upload(data)
        co = scan.chunk_offsets(data)
        best = 1e9
        for rep in range(4):
            s.reset()
            t0 = time.perf_counter()
            s.scan_device(d.ptr, data.nbytes, co)
            r = s.finish()
            best = min(best, time.perf_counter() - t0)
        times[name] = best
        if name == 'long':
            assert r['path'] == REDO_PATH, r['path']
            assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
            assert r['coverage'].tolist() == o['coverage'] and r['mutations'].tolist() == o['mutations']
            assert r['stats']['records_parsed'] == n + 1 and r['stats']['readlengths'] == o['stats']['readlengths']
        else:
            assert r['path'] == dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False), r['path']
        d.free()
    # (the redo costs a round trip to the host and a few small kernels; a rescan of the batch with the
    # exhaustive kernels took about 80 ms for these 300 k reads)
    assert times['long'] < 2 * times['plain'] + 0.004, times
    s.close(); t.close()


def test_long_records_with_ragged_quality_go_through_the_redo_exactly():
    """a dozen records of 1.1 .. 9 kB that outgrow their tiles' look-ahead, their score lines full of dips, ties between
    equally long good runs and runs that cross the 16-byte, 1 KiB and 4 KiB seams of the wave-parallel trim: hits, read
    lengths and counters equal the oracle's (the redo of skipped tiles: kvq_collect_skipped, kvq_trim_records'
    long-line path, the matcher's launch for long reads)"""
    import random
    rng = random.Random(20261004)
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    n, L = 120000, 150
    rb = synth.record_bytes(L)
    plain = synth.reads(g, 0, n, L)
    co_plain = scan.chunk_offsets(plain)
    pieces, at = [], 0
    lens = [1100, 2047, 2048, 2049, 4095, 4096, 4097, 5000, 6000, 8191, 9000, 1025]
    tabs = synth.table(g)
    for i, ln in enumerate(lens):
        c = 1 + i * ((len(co_plain) - 2) // len(lens))
        cut = int(co_plain[c]) + 108 * rb                          # (inside the chunk's first tile, the record's end beyond its look-ahead)
        st = rng.randrange(0, len(g) - ln - 1)
        bases = bytes(g[st:st + ln])
        q = bytearray(b'I' * ln)
        kind = i % 4
        if kind == 0:                                              # dips everywhere
            for _ in range(ln // 40): q[rng.randrange(ln)] = ord('#')
        elif kind == 1:                                            # two equally long best runs: the first one wins
            run = ln // 3
            q[:] = b'#' * ln
            a = rng.randrange(1, ln - 2 * run - 2); b = a + run + 1 + rng.randrange(0, ln - a - 2 * run - 1)
            q[a:a + run] = b'I' * run; q[b:b + run] = b'I' * run
        elif kind == 2:                                            # the best run ends with the line (closed by its newline)
            for _ in range(ln // 100): q[rng.randrange(ln // 2)] = ord('#')
        else:                                                      # short runs, the longest somewhere across a 1 KiB seam
            q[:] = bytes(rng.choice(b'#I') for _ in range(ln))
            a = 1024 * (1 + rng.randrange(max(1, ln // 1024 - 1))) - 37
            q[a:a + 90] = b'I' * 90; q[a - 1:a] = b'#'; q[a + 90:a + 91] = b'#'
        pieces.append(plain[at:cut].tobytes()); at = cut
        pieces.append(b'@long%d 1:N:0\n' % i + bases + b'\n+\n' + bytes(q) + b'\n')
    pieces.append(plain[at:].tobytes())
    text = np.frombuffer(b''.join(pieces), dtype=np.uint8)
    cfg = dict(cases.PRODUCT)
    o = O.scan_memory(text, seqs, fold=True, **dict(cfg, nthreads=16))
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(text.nbytes); d.upload(text)
    co = scan.chunk_offsets(text)
    for rep in range(2):                                           # (the second time the launches for a scan that has seen skipped tiles)
        s.reset()
        s.scan_device(d.ptr, text.nbytes, co)
        r = s.finish()
        assert r['path'] == REDO_PATH, r['path']
        assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
        assert r['coverage'].tolist() == o['coverage'] and r['mutations'].tolist() == o['mutations']
        assert r['stats']['records_parsed'] == n + len(lens) and r['stats']['readlengths'] == o['stats']['readlengths']
        assert r['stats']['nseqhits'] == o['stats']['nseqhits'] and r['stats']['nseqbasehits'] == o['stats']['nseqbasehits']
    d.free(); s.close(); t.close()


def test_a_read_that_floods_its_waves_queues_is_matched_on_its_own():
    """300 sequences that share a repeat: a read made of that repeat finds thousands of index entries, more than a wave's
    queues hold even when it is the wave's only read.  Such a read goes to the exhaustive matcher on its own (the redo's
    list, filled by the scan kernel), its tile and its batch carry on: same hits as the oracle, no rescan of the batch"""
    import random
    rng = random.Random(7)
    unit = 'ACGTTGCA'
    seqs = [unit * 5 + ''.join(rng.choice('ACGT') for _ in range(20)) for _ in range(300)]
    g = synth.genome()
    n, L = 20000, 150
    rb = synth.record_bytes(L)
    plain = synth.reads(g, 0, n, L)
    flood = [(unit * 19)[:150], (unit * 19)[3:150], (unit * 200)[:1500], (unit * 19)[:150]]      # (one of them long: the long reads' launch)
    pieces, at = [], 0
    for i, b in enumerate(flood):
        cut = (1 + i * 4500) * rb
        pieces.append(plain[at:cut].tobytes()); at = cut
        pieces.append(('@flood%d\n%s\n+\n%s\n' % (i, b, 'I' * len(b))).encode())
    pieces.append(plain[at:].tobytes())
    text = np.frombuffer(b''.join(pieces), dtype=np.uint8)
    cfg = dict(cases.PRODUCT)
    o = O.scan_memory(text, seqs, fold=True, **dict(cfg, nthreads=16))
    assert len(o['hits']) > 1000
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    d = scan.DeviceBuffer(text.nbytes); d.upload(text)
    co = scan.chunk_offsets(text)
    for rep in range(2):
        s.reset()
        s.scan_device(d.ptr, text.nbytes, co)
        r = s.finish()
        if os.environ.get('KVQ_DENSE') == '1' or os.environ.get('KVQ_K') in ('5', '6', '7'):   # (the draining kernels -- always, for seeds shorter than 8 -- take such a read themselves: nothing is left for the redo)
            assert r['path'] == dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False), r['path']
        else:
            assert r['path'] == REDO_PATH, r['path']               # (records behind the scan, not the batch again)
        assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
        assert r['coverage'].tolist() == o['coverage'] and r['mutations'].tolist() == o['mutations']
        assert r['stats']['records_parsed'] == n + len(flood) and r['stats']['readlengths'] == o['stats']['readlengths']
        assert r['stats']['nseqhits'] == o['stats']['nseqhits'] and r['stats']['nseqbasehits'] == o['stats']['nseqbasehits']
    d.free(); s.close(); t.close()


def test_reads_with_more_work_items_than_a_queue_holds_are_dealt_with_in_turns():
    """a dense table -- 64 variants of each of four templates -- makes every read off a template meet thousands of
    (candidate, index entry) pairs, many times what a wave's queue holds: the queue is filled and emptied in turns
    (no redo, no rescan) and the hits equal the oracle's"""
    import random
    rng = random.Random(11)
    g = synth.genome()
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    seqs, spots = [], [3000, 9000, 15000, 21000]
    for a in spots:
        base = bytes(g[a:a + 120])
        for v in range(64):
            b = bytearray(base)
            b[60 + (v % 50)] = ord('ACGT'[(('ACGT'.index(chr(b[60 + (v % 50)]))) + 1 + v // 50) % 4])      # one changed base per variant
            seqs.append(bytes(b[v % 7: 100 + v % 13]).decode())
    n, L = 6000, 150
    rb = synth.record_bytes(L)
    plain = synth.reads(g, 0, n, L).tobytes()
    recs = []
    for i in range(400):                                             # reads off the templates, both strands, a few errors
        a = rng.choice(spots) + rng.randrange(-60, 40)
        b = bytearray(g[a:a + L])
        for _ in range(rng.randrange(0, 3)): b[rng.randrange(L)] = ord(rng.choice('ACGT'))
        if i & 1: b = bytearray(comp[c] for c in reversed(b))
        recs.append(b'@dense%d\n' % i + bytes(b) + b'\n+\n' + b'I' * L + b'\n')
    text = np.frombuffer(plain[:3000 * rb] + b''.join(recs) + plain[3000 * rb:], dtype=np.uint8)
    cfg = dict(cases.PRODUCT)
    o = O.scan_memory(text, seqs, fold=True, **dict(cfg, nthreads=16))
    assert len(o['hits']) > 10000
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    r = s.scan_host(text, scan.chunk_offsets(text)) or s.finish()
    assert r['path'] == dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False), r['path']
    assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
    assert r['coverage'].tolist() == o['coverage'] and r['mutations'].tolist() == o['mutations']
    assert r['stats']['nseqhits'] == o['stats']['nseqhits'] and r['stats']['nseqbasehits'] == o['stats']['nseqbasehits']
    s.close(); t.close()


def test_speculation_failure_falls_back_to_the_exact_split():
    """a FastQ whose base lines may start with '@' or '+' defeats the text heuristic;
    the validation pass must notice and the rescan must give the reference's answer"""
    import random
    rng = random.Random(99)
    target = cases.QUIRK_SEQ
    recs = []
    for i in range(4000):
        bases = cases.randseq(rng, rng.randint(60, 200))
        if i % 7 == 0:
            bases = '@' + bases[1:]
        if i % 11 == 0:
            bases = '+' + bases[1:]
        if i % 5 == 0:
            at = rng.randint(1, len(bases) - 1)
            bases = (bases[:at] + target)[:230]
        q = ''.join(rng.choice('@+IIII') for _ in bases)
        recs.append(cases.rec('r%d' % i, bases, q))
    data = np.frombuffer(b''.join(recs), dtype=np.uint8)
    seqs = synth.both_strands([target.encode()])
    cfg = dict(cases.PRODUCT, Amin='!')
    o = O.scan_memory(data, seqs, fold=True, **dict(cfg, nthreads=4))
    t = scan.Table(seqs, **cfg)
    s = scan.Scanner(t)
    s.scan_host(data)
    r = s.finish()
    assert tuple(r['hits']) == tuple(o['hits']) and r['hitseqs'] == o['hitseqs']
    assert r['stats']['readlengths'] == o['stats']['readlengths'] and r['stats']['records_parsed'] == 4000
    assert r['coverage'].tolist() == o['coverage']
    assert len(o['hits']) > 100
    s.close(); t.close()


def test_dense_table_overflows_the_candidate_queues_gracefully():
    """8x the MTBC table (2112 sequences): the 8-mer bitmap is dense, the candidate queues of a
    tile overflow and its reads are filtered in smaller stretches -- same hits as the exhaustive
    kernels, without a rescan"""
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g, 'MTBC', scale=8))
    data = synth.reads(g, 777, 20000, 150)
    t = scan.Table(seqs, **cases.PRODUCT)
    res = []
    for force in (False, True):
        s = scan.Scanner(t)
        s.force_exhaustive(force)
        s.scan_host(data)
        res.append(s.finish())
        s.close()
    a, b = res
    assert a['path'] == dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False)
    assert a['hits'] == b['hits'] and a['hitseqs'] == b['hitseqs']
    assert (a['counters'] == b['counters']).all()
    assert len(a['hits']) > 300
    t.close()


def test_reads_longer_than_4095_bases_keep_their_late_candidates(tmp_path):
    """a read may fill a tile's window (tens of kilobases): candidate positions beyond 4095 must survive the
    queue (they were once kept in twelve bits: hits deep in a long read were lost, others reported twice).
    Found by tools/fuzz_parity.py, seeds 20325, 20667, 20774, 21304 of the seeded campaign."""
    import random
    rnd = random.Random(5)
    genome = ''.join(rnd.choice('ACGT') for _ in range(12000))
    recs = []
    for i, (st, L) in enumerate([(100, 4200), (3000, 9000), (0, 4096), (2000, 4097), (500, 6000), (7000, 150)]):
        recs.append('@r%d\n%s\n+\n%s\n' % (i, genome[st:st + L], 'I' * L))
    data = ''.join(recs).encode()
    seqs = [genome[a:a + n].encode() for a, n in ((100 + 4175, 25), (100 + 79, 25), (3000 + 8344, 25), (3000 + 8087, 39),
                                                  (4090, 25), (6050, 51), (6400, 25), (11000, 100), (2000 + 4072, 25))]
    p = tmp_path / 'long.fastq'
    p.write_bytes(data)
    cfg = dict(cases.PRODUCT, maxerrors=2, minoverlap=25, minreadlength=25, Amin='#', nthreads=1)
    engine.config(**cfg)
    r = engine.findseqs(str(p), seqs)
    o = O.findseqs(str(p), seqs, **cfg)
    assert len(o['hits']) >= 12 and max(-h.seq_pos for h in o['hits']) > 8000
    assert tuple(r['hits']) == tuple(o['hits'])
    assert [bytes(h) for h in r['hitseqs']] == o['hitseqs'] and r['stats'] == o['stats']
    # the records that fit one tile's window together, as a device batch: the seed-filter kernel itself took
    # these reads (no tile was handed to the exhaustive kernels)
    data = ''.join(recs[:3] + recs[5:]).encode()
    (tmp_path / 'fit.fastq').write_bytes(data)
    o = O.findseqs(str(tmp_path / 'fit.fastq'), seqs, **cfg)
    assert max(-h.seq_pos for h in o['hits']) > 8000
    t = scan.Table(seqs, **{k: v for k, v in cfg.items() if k != 'nthreads'})
    s = scan.Scanner(t)
    arr = np.frombuffer(data, dtype=np.uint8)
    d = scan.DeviceBuffer(arr.nbytes); d.upload(arr)
    s.scan_device(d.ptr, arr.nbytes, scan.chunk_offsets(arr))
    q = s.finish()
    if os.environ.get('KVQ_K', '8') in ('', '0', '6', '7', '8'):   # (a 5-mer filter lets a 9 000-base read flood its wave's queue: that read goes through the redo, by design)
        assert q['path'] == dict(seeded=True, exhaustive=False, rescanned=False, tiles_rescanned=False), q['path']
    assert tuple(q['hits']) == tuple(o['hits'])
    s.close(); t.close(); d.free()


@pytest.mark.parametrize('malformed', [False, True])
def test_a_tile_that_keeps_some_records_still_answers_for_its_first_one(tmp_path, malformed):
    """long records: a tile whose last record outgrows its window keeps the records in front of it and leaves the
    rest to the exhaustive kernels -- so its speculated first record counts and must be validated like any tile's.
    (tools/fuzz_parity.py seed 131412: the first record of such a tile had a malformed '+' line, the speculation
    skipped it, and nobody reported the error.)"""
    import random
    rnd = random.Random(7)
    genome = ''.join(rnd.choice('ACGT') for _ in range(20000))
    lens = [4200, 9000, 9000, 1024, 1023, 4200, 4200, 4200, 4200, 1500, 1500, 4200, 9000, 600, 1023, 9000, 600]
    recs, at = [], 0
    for i, L in enumerate(lens):
        st = rnd.randrange(0, len(genome) - L)
        recs.append('@r%d\n%s\n+\n%s\n' % (i, genome[st:st + L], 'I' * L))
    if malformed:
        recs[8] = recs[8].replace('\n+\n', '\n-\n', 1)
    data = ''.join(recs).encode()
    starts = [sum(len(r) for r in recs[:i]) for i in range(len(recs))]
    assert 73280 < starts[8] < 77440                        # the record sits at the head of the third tile (tiles own 36640 bytes here)
    seqs = [genome[a:a + n].encode() for a, n in ((100, 150), (5000, 51), (12000, 25), (15000, 300))]
    p = tmp_path / 'partial.fastq'
    p.write_bytes(data)
    cfg = dict(cases.PRODUCT, maxerrors=2, minoverlap=25, minreadlength=25, Amin='#', nthreads=1)
    engine.config(**cfg)
    if malformed:
        with pytest.raises(O.OracleFormatError) as eo:
            O.findseqs(str(p), seqs, **cfg)
        with pytest.raises(FastqFileFormatException) as eg:
            engine.findseqs(str(p), seqs)
        assert str(eg.value) == str(eo.value) and 'fpos=%d' % (starts[8] + len('@r8\n') + 4200 + 1) in str(eg.value)
        return
    o = O.findseqs(str(p), seqs, **cfg)
    r = engine.findseqs(str(p), seqs)
    assert len(o['hits']) > 0 and tuple(r['hits']) == tuple(o['hits'])
    assert [bytes(h) for h in r['hitseqs']] == o['hitseqs'] and r['stats'] == o['stats']
    t = scan.Table(seqs, **{k: v for k, v in cfg.items() if k != 'nthreads'})
    s = scan.Scanner(t)
    arr = np.frombuffer(data, dtype=np.uint8)
    d = scan.DeviceBuffer(arr.nbytes); d.upload(arr)
    s.scan_device(d.ptr, arr.nbytes, scan.chunk_offsets(arr))
    q = s.finish()
    assert q['path'] == REDO_PATH, q['path']
    assert tuple(q['hits']) == tuple(o['hits']) and q['stats']['records_parsed'] == len(lens)
    s.close(); t.close(); d.free()
