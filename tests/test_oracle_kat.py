"""
The oracle restatement against the LITERAL expectations of the reference's own
tests (tests/test_engine.py of the reference; line numbers in each test).  These
need neither oracle/_ref nor expected.json.
"""
import gzip
import os

import pytest

import cases
from oracle import oracle as O


def F(fastqs, name):
    return os.path.join(fastqs, name)


def check_findseqs(fname, opener):
    # test_engine.py:137-175
    seqs = cases.SEQS_FINDSEQS
    hits = O.findseqs(fname, seqs, maxerrors=0, minoverlap=1000, minreadlength=3, Amin='!')['hits']
    data = opener(fname).read()
    x = [0] * len(seqs)
    for hit in hits:
        x[hit.seq_nr] += 1
        seq = seqs[hit.seq_nr].encode()
        if hit.seq_pos < 0:
            bps = data[hit.file_pos - hit.seq_pos:hit.file_pos - hit.seq_pos + hit.length]
        else:
            bps = data[hit.file_pos:hit.file_pos + hit.length]
            seq = seq[hit.seq_pos:hit.seq_pos + hit.length]
        assert bps == seq
    assert x == [19, 1, 0, 1, 1, 1, 1]


def test_findseqs(fastqs):
    check_findseqs(F(fastqs, 'test_engine.fastq'), lambda f: open(f, 'rb'))


def test_gz(fastqs):
    # test_engine.py:178-181
    check_findseqs(F(fastqs, 'test_engine.fastq.gz'), lambda f: gzip.GzipFile(f, 'rb'))


@pytest.mark.parametrize('gz', ['', '.gz'])
def test_paired(fastqs, gz):
    # test_engine.py:184-205: whole result dicts are equal
    cfg = dict(maxerrors=0, minoverlap=1000, minreadlength=3, Amin='!')
    ret = O.findseqs(F(fastqs, 'test_engine.fastq' + gz), cases.SEQS_FINDSEQS, **cfg)
    ret_12 = O.findseqs((F(fastqs, 'test_engine_1.fastq' + gz), F(fastqs, 'test_engine_2.fastq' + gz)), cases.SEQS_FINDSEQS, **cfg)
    assert ret == ret_12


def test_maxerror(fastqs):
    # test_engine.py:208-224
    for maxerrors in range(4):
        hits = O.findseqs(F(fastqs, 'test_engine.fastq'), cases.SEQS_MAXERROR, minreadlength=25, minoverlap=25,
                          Amin='!', maxerrors=maxerrors)['hits']
        assert len(hits) == maxerrors


def test_minoverlap(fastqs):
    # test_engine.py:227-254
    f = F(fastqs, 'test_engine.fastq')
    hits = O.findseqs(f, cases.SEQS_MINOVERLAP, maxerrors=0, minreadlength=25, minoverlap=30, Amin='!')['hits']
    assert len(hits) == 1 and hits[0].seq_nr == 0 and hits[0].seq_pos < 0
    hits = O.findseqs(f, cases.SEQS_MINOVERLAP, maxerrors=0, minreadlength=25, minoverlap=25, Amin='!')['hits']
    assert len(hits) == 2
    for hit in hits:
        assert hit[0] != 3 or hit[2] > 0
    hits = O.findseqs(f, cases.SEQS_MINOVERLAP, maxerrors=1, minreadlength=25, minoverlap=25, Amin='!')['hits']
    assert len(hits) == 4


def test_Amin(fastqs):
    # test_engine.py:257-271
    f = F(fastqs, 'test_engine.fastq')
    ret = O.findseqs(f, cases.SEQS_AMIN, Amin='H', minreadlength=4, maxerrors=0, minoverlap=25)
    assert len(ret['hits']) == 1
    assert ret['stats']['readlengths'][5] == 3
    assert ret['stats']['readlengths'][4] == 5
    ret = O.findseqs(f, cases.SEQS_AMIN, Amin='G', minreadlength=4, maxerrors=0, minoverlap=25)
    assert len(ret['hits']) == 2


def test_hits(tmp_path):
    # test_engine.py:274-321 (generator restated in cases.cover_file)
    p = tmp_path / 'cover.fastq'
    p.write_bytes(cases.COVER_BYTES)
    cfg = dict(nthreads=3, Amin='5', maxerrors=0, minreadlength=60, minoverlap=25)
    ret = O.findseqs(str(p), [cases.COVER_SEQ], **cfg)
    assert ret['stats']['readlengths'][100] == 100
    assert len(ret['hits']) == 100
    ret = O.findseqs(str(p), [cases.COVER_SEQX], **cfg)
    assert ret['stats']['readlengths'][100] == 100
    assert len(ret['hits']) == 0


def test_fastq_format_errors(tmp_path):
    # test_engine.py:324-346
    for name in ('bad_at', 'bad_plus'):
        c = cases.by_name(big=False)[name]
        with pytest.raises(O.OracleFormatError):
            O.findseqs(c.materialize(tmp_path)[0], [], **c.config)


def test_forward_fastq(tmp_path):
    # test_engine.py:349-359
    for n in (3, 5, 7, 133):
        for plus in ('+', '+IDENTIFIER'):
            for nl in ('\n', '\r\n'):
                p = tmp_path / 'fw.fastq'
                p.write_bytes(cases.forward_file(n, plus, nl))
                ret = O.findseqs(str(p), ['A' * 80], Amin='#', nthreads=2, minoverlap=80, maxerrors=2, minreadlength=25)
                assert len(ret['hits']) == n


def test_missing_file_is_an_ioerror(tmp_path):
    # intended behaviour of workhorse.c:661-668 (the reference itself has a use-after-free there)
    with pytest.raises(IOError) as ei:
        O.findseqs(str(tmp_path / 'nope.fastq'), ['ACGT'])
    assert 'for getting filesize' in str(ei.value)


def test_survey_appendix_b_values(fastqs):
    # SURVEY.md Appendix B (observed from the reference engine)
    r = O.findseqs(F(fastqs, 'test_engine.fastq'), cases.SEQS_FINDSEQS, maxerrors=0, minoverlap=1000, minreadlength=3, Amin='!')
    assert r['stats']['nseqbasehits'] == (57, 4, 0, 5, 5, 50, 51)
    assert r['stats']['readlengths'][51] == 14 and len(r['stats']['readlengths']) == 52
    assert r['stats']['parsed'] == r['stats']['total'] == 2240 and r['stats']['records_parsed'] == 14
    assert r['hits'][:3] == (O.Hit(0, 54, -12, 3, 51), O.Hit(1, 54, -20, 4, 51), O.Hit(3, 54, -1, 5, 51))
    assert r['hitseqs'][:3] == [b'CCC', b'TTTT', b'TGTAG']
    r = O.findseqs(F(fastqs, 'test_engine.fastq'), cases.SEQS_MINOVERLAP, maxerrors=0, minreadlength=25, minoverlap=25, Amin='!')
    assert r['hits'] == (O.Hit(0, 694, -21, 30, 51), O.Hit(2, 1174, 3, 25, 51))
    c = cases.by_name(big=False)['quirk']
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        r = O.findseqs(c.materialize(t)[0], c.seqs, **c.config)
    got = [(h.seq_pos, h.length, h.readlength) for h in r['hits']]
    assert got == [(21, 30, 30), (21, 30, 30), (0, 30, 30), (10, 30, 30), (0, 51, 51), (-5, 51, 61), (-8, 30, 38), (21, 30, 38)]
    r = O.findseqs(F(fastqs, 'test_analyser.fastq'), cases.SPOLIGO, minoverlap=10, maxerrors=1, minreadlength=10, Amin='!')
    assert r['stats']['records_parsed'] == 72 and len(r['hits']) == 4
    assert r['stats']['nseqhits'][0] == 2 and r['stats']['nseqhits'][42] == 2
    with pytest.raises(O.OracleFormatError) as ei:
        O.findseqs(F(fastqs, 'L3_N1014_hits_500_BROKEN.fastq'), cases.SPOLIGO, **cases.PRODUCT)
    assert str(ei.value) == "record must start with '@' (and not '.') fpos=18534"


def test_coverage_fold_matches_test_analyser():
    # test_analyser.py:71-107: template AACCGGTT, three hand-written hits
    import ctypes as C
    seq = b'AACCGGTT'
    hits = [(0, 8, b'ATCCGGTTTT'), (-2, 8, b'AACCGGTT'), (-1, 8, b'ATCCGGTTA')]
    L = O.lib()
    r = O.Result()
    n = len(hits)
    seq_nr = (C.c_int32 * n)(*[0] * n)
    seq_pos = (C.c_int32 * n)(*[h[0] for h in hits])
    length = (C.c_int32 * n)(*[h[1] for h in hits])
    blob = b''.join(h[2][:h[1]] for h in hits)
    off = (C.c_int64 * (n + 1))(*[8 * i for i in range(n + 1)])
    bb = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
    r.n_hits = n
    r.seq_nr = seq_nr
    r.seq_pos = seq_pos
    r.length = length
    r.hitseq_blob = bb
    r.hitseq_off = off
    sarr = (C.c_char_p * 1)(seq)
    lens = (C.c_int32 * 1)(8)
    soff = (C.c_int64 * 2)(0, 8)
    cov = (C.c_int64 * 8)()
    mut = (C.c_int64 * 48)()
    L.kvo_fold_coverage(C.byref(r), sarr, lens, soff, cov, mut)
    assert list(cov) == [3] * 8
    m = [list(mut)[i * 6:(i + 1) * 6] for i in range(8)]
    assert m[1] == [0, 0, 0, 2, 0, 0]          # two reads carry T at index 1 (fractions T 2/3, A 1/3)
    assert all(sum(m[i]) == 0 for i in range(8) if i != 1)
