"""host-side callers of the engine: the ``Fastq`` probe (reference tests/test_fastq.py) and the
coverage / JSON round trip of the scan-level ``Analyser`` (reference tests/test_analyser.py:30-107).
The inputs and expected outcomes are the reference tests'; the format cases run on plain and gzipped files."""
import gzip
import io
import json
import logging

import pytest

from kvarq_amd import analyse, engine
from kvarq_amd.coverage import Coverage, Sequence
from kvarq_amd.engine import Hit
from kvarq_amd.fastq import Fastq, FastqFileFormatException


def write_fastq(tmp_path, content, gz, variant=None, **kw):
    p = tmp_path / ('t.fastq.gz' if gz else 't.fastq')
    (gzip.open(str(p), 'wb') if gz else open(str(p), 'wb')).write(content.encode('latin-1'))
    return Fastq(str(p), variant=variant, quiet=True, **kw)


def write_quality(tmp_path, quality, gz, variant=None):
    return write_fastq(tmp_path, '@IDENTIFIER\n' + 'A' * len(quality) + '\n+\n' + quality + '\n', gz, variant)


@pytest.mark.parametrize('gz', [False, True])
def test_fastq_variant(tmp_path, gz, caplog):
    # test_fastq.py:47-81
    fq = write_quality(tmp_path, '!"#$%&\'()*+,-./0123456789:;<=>?@ABCDEFGHIJ', gz)
    assert fq.dQ == 0 and set(fq.variants) == {'Illumina 1.8+', 'Sanger'} and fq.Azero == '!'
    fq = write_quality(tmp_path, ';<=>?@ABCDEFGHIJKLMNOPQRSTUVWXYZ[\\]^_`abcdefgh', gz)
    assert fq.dQ == 31 and fq.variants == ['Solexa']
    fq = write_quality(tmp_path, 'OPQRSTUVWXYZ[\\]^_`abcdefgh', gz)
    assert fq.dQ == 31 and fq.variants == ['Solexa', 'Illumina 1.3+', 'Illumina 1.5+'] and fq.Azero == '@'
    with pytest.raises(FastqFileFormatException):
        write_quality(tmp_path, ';<=>?@ABCDEFGHI;<=>?@ABCDEFGHI', gz)               # ambiguous
    write_quality(tmp_path, ';<=>?@ABCDEFGHI;<=>?@ABCDEFGHI', gz, variant='Sanger')
    write_quality(tmp_path, ';<=>?@ABCDEFGHI;<=>?@ABCDEFGHI', gz, variant='Solexa')
    with caplog.at_level(logging.WARNING, logger='kvarq'):
        write_quality(tmp_path, ';<=>?@ABCDEFGHI;<=>?@ABCDEFGHI', gz, variant='Illumina 1.3+')
    assert any('seems not to be compatible' in r.getMessage() for r in caplog.records)
    with pytest.raises(FastqFileFormatException):
        write_quality(tmp_path, 'IIII', gz, variant='no such vendor')


BASES = 'ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT'

BAD_FILES = [
    'IDENTIFIER\n' + BASES + '\n+\n' + '#' * 44 + '\n',                              # identifier without '@'
    '@IDENTIFIER\n' + BASES.replace('CGTACGTACGTACGTACGTAC', 'CGTACGTACGTACGTACGTAX', 1) + '\n+\n' + '#' * 44 + '\n',   # 'X' among the bases
    '@IDENTIFIER\n' + BASES + '\n+\n' + '#' * 45 + '\n',                              # scores longer than bases
    '@IDENTIFIER\n' + BASES + '\n+text\n' + '#' * 44 + '\n',                          # '+text' that is not the identifier
    '@IDENTIFIER\n' + BASES + '\n+\n' + '#' * 44 + '\n\n@IDENTIFIER\n',               # text behind an empty line
]


@pytest.mark.parametrize('gz', [False, True])
@pytest.mark.parametrize('content', BAD_FILES)
def test_fastq_format(tmp_path, gz, content):
    # test_fastq.py:83-135
    with pytest.raises(FastqFileFormatException):
        write_fastq(tmp_path, content, gz)


def test_fastq_accepts_what_the_reference_accepts(tmp_path):
    ok = ('@ID 1\n' + BASES + '\n+ID 1\n' + 'I' * 44 + '\n' +                        # '+' followed by the identifier
          '@ID 2\n' + BASES + '\n+\n' + 'I' * 44 + '!\n' +                            # one trailing '!' more than bases
          '\n\n')                                                                    # empty lines at the end
    fq = write_fastq(tmp_path, ok, False)
    assert fq.readlength == 44 and fq.dQ == 0 and fq.records_approx == len(ok) // len('@ID 1\n' + BASES + '\n+ID 1\n' + 'I' * 44 + '\n')
    with pytest.raises(FastqFileFormatException):
        Fastq(str(tmp_path / 'reads.txt'))
    (tmp_path / 'empty.fastq').write_bytes(b'')
    with pytest.raises(FastqFileFormatException):
        Fastq(str(tmp_path / 'empty.fastq'))


def test_fastq_pairs_phred_and_records(tmp_path):
    rec = lambda i, q='#' + 'I' * 19: '@r%03d\n%s\n+\n%s\n' % (i, 'ACGTACGTACGTACGTACGT', q)
    text = ''.join(rec(i) for i in range(500))
    (tmp_path / 's_1.fastq').write_text(text)
    (tmp_path / 's_2.fastq').write_text(text)
    fq = Fastq(str(tmp_path / 's_1.fastq'), paired=True, quiet=True)
    assert fq.filenames() == [str(tmp_path / 's_1.fastq'), str(tmp_path / 's_2.fastq')]
    assert fq.filesizes() == [len(text)] * 2 and fq.records_approx == 1000 and fq.readlength == 20
    assert Fastq(str(tmp_path / 's_1.fastq'), quiet=True).filenames() == [str(tmp_path / 's_1.fastq')]
    # PHRED arithmetic (fastq.py:238-263): Q13 on Sanger/Illumina 1.8+ is the product's Amin '.'
    assert fq.Q2A(13) == '.' and fq.A2Q('.') == 13 and fq.Q2A(0) == '!' and abs(fq.Q2p(20) - 0.01) < 1e-12 and fq.p2Q(0.001) == 29 and fq.p2Q(0.0009) == 30      # int() truncates, as in the reference
    # cutoff: longest CLOSED run of good scores (fastq.py:295-308)
    assert Fastq.cutoff('II#IIII#I', '.') == (3, 4) and Fastq.cutoff('IIII', '.') == (0, -1) and Fastq.cutoff('##', '.') == (0, 0)
    assert fq.lengths('.', n=50) == [0] * 50          # '#' closes an empty run; the run of 'I' up to the line end is not closed
    # seekback / readrecordat / readhit: any position inside record 7's score or '+' line -> record 7; inside its
    # identifier or bases -> record 6
    start7 = 7 * len(rec(0))
    for off, want in ((0, 6), (3, 6), (10, 6), (len('@r007\n') + 20 + 1, 7), (len(rec(7)) - 2, 7)):
        fq.fd.seek(start7 + off); fq.seekback()
        assert fq.readrecord()[0] == '@r%03d' % want, (off, want)
    fq.fd.seek(5); fq.seekback(); assert fq.fd.tell() == 0
    hit = Hit(seq_nr=0, file_pos=start7 + len('@r007\n') + 4, seq_pos=-4, length=8, readlength=20)
    assert fq.readhit(hit) == 'ACGTACGT'[0:8] and fq.readhits([hit]) == ['ACGTACGT']
    assert fq.readrecordat(hit) == rec(7)
    # a score line that starts with '+' must not be taken for the separator
    tricky = rec(0) + '@r001\nACGTACGTACGTACGTACGT\n+\n+IIIIIIIIIIIIIIIIII#\n' + rec(2)
    (tmp_path / 'tricky.fastq').write_text(tricky)
    t = Fastq(str(tmp_path / 'tricky.fastq'), quiet=True)
    t.fd.seek(len(rec(0)) + len('@r001\nACGTACGTACGTACGTACGT\n+\n') + 5); t.seekback()
    assert t.readrecord()[0] == '@r001'


def test_coverage_fold():
    # test_analyser.py:71-107
    #   AACCGGTT    : template
    #   ATCCGGTTTT  : hit1
    # AAAACCGGTT    : hit2
    #  AATCCGGTTA   : hit3
    cov = Coverage(Sequence('AACCGGTT'))
    cov.apply_hit(Hit(seq_nr=0, file_pos=-1, seq_pos=0, length=8, readlength=10), 'ATCCGGTTTT', on_plus_strand=True)
    assert cov.minf() == 1 and not cov.mixed() and tuple(cov.coverage) == (1,) * 8 and 1 in cov.mutations
    cov.deserialize(cov.serialize())
    assert tuple(cov.coverage) == (1,) * 8 and 1 in cov.mutations
    cov.apply_hit(Hit(seq_nr=0, file_pos=-1, seq_pos=-2, length=8, readlength=10), 'AACCGGTT', on_plus_strand=True)
    cov.apply_hit(Hit(seq_nr=0, file_pos=-1, seq_pos=-1, length=8, readlength=10), 'ATCCGGTTA', on_plus_strand=True)
    assert 0.65 < cov.minf() < 0.69 and cov.mixed()
    fs = cov.fractions_at(1)
    assert list(fs.keys())[0] == 'T' and list(fs.values())[0] > 0.65 and list(fs.keys())[1] == 'A' and list(fs.values())[1] < 0.35


def test_json_dump_layout_and_round_trip():
    data = dict(analyses={}, info={'format': 'kvarq', 'fastq': ['a_1.fastq', 'a_2.fastq'], 'config': {'Amin': '.', 'maxerrors': 2}},
                stats={'readlengths': [0, 1, 2], 'progress': 1.0}, coverages=[('SNP1', '1-2-3 1[T]'), ('SNP2', '0-0 ')],
                hits=[[0, 10, -2, 8, 10]])
    out = io.StringIO()
    analyse.json_dump(data, out)
    text = out.getvalue()
    assert json.loads(text) == json.loads(json.dumps(data))
    lines = text.split('\n')
    assert '  "coverages": [' in lines and '    ["SNP1", "1-2-3 1[T]"], ' in lines         # third level on one line (", " ends the line: Python 2.7's separator)
    assert '    "fastq": ["a_1.fastq", "a_2.fastq"], ' in lines and '    "readlengths": [0, 1, 2], ' in lines
    assert lines[0] == '{' and lines[-1] == '}'


@pytest.mark.gpu
def test_analyser_scan_encode_decode(tmp_path):
    """scan -> encode(hits=True) -> JSON text -> decode -> same coverages (test_analyser.py:30-49), and the
    per-hit fold equals the device fold of the same scan"""
    from kvarq_amd import scan as kscan, synth
    from kvarq_amd.coverage import coverages_from_scan
    import numpy as np
    g = synth.genome()
    plus = synth.table(g)[:40]
    n, L = 20000, 150
    reads = synth.reads(g, 0, n, L)
    p = tmp_path / 'sample_1.fastq'
    p.write_bytes(reads.tobytes())
    (tmp_path / 'sample_2.fastq').write_bytes(synth.reads(g, n, 5000, L).tobytes())
    engine.config(maxerrors=2, minoverlap=25, minreadlength=25, Amin='.', nthreads=2)
    templates = dict(('T%03d' % i, (s.decode(), 25, 25) if len(s) == 51 else s.decode()) for i, s in enumerate(plus))
    fq = Fastq(str(p), paired=True, quiet=True)
    assert fq.Q2A(13) == '.' and fq.readlength == L and len(fq.filenames()) == 2
    a = analyse.Analyser()
    a.scan(fq, templates)
    assert a.stats['records_parsed'] == n + 5000 and len(a.hits) > 0 and a[0] is a['T000'] and a[len(a)] is a['T000']
    out = io.StringIO()
    analyse.json_dump(a.encode(hits=True), out)
    data = json.loads(out.getvalue())
    assert data['info']['fastq'] == fq.filenames() and data['info']['config']['Amin'] == '.' and data['info']['format'] == 'kvarq'
    b = analyse.Analyser()
    b.decode(templates, data)
    assert [c.serialize() for c in b.coverages.values()] == [c.serialize() for c in a.coverages.values()]
    assert b.hits == list(a.hits) and b.stats['nseqhits'] == list(a.stats['nseqhits'])
    b.update_coverages()
    assert [c.serialize() for c in b.coverages.values()] == [c.serialize() for c in a.coverages.values()]
    assert b['T003'].start == 25 and b['T003'].stop == 26
    with pytest.raises(analyse.DecodingException):
        b.decode(templates, {'info': {'format': 'other'}})
    # the device fold of the same text gives the same coverages without the per-hit loop
    both = synth.both_strands(plus)
    t = kscan.Table(both, maxerrors=2, minoverlap=25, minreadlength=25, Amin='.')
    s = kscan.Scanner(t)
    allreads = np.concatenate([reads, synth.reads(g, n, 5000, L)])
    s.scan_host(allreads, kscan.chunk_offsets(allreads))
    r = s.finish(hits=False)
    dev = coverages_from_scan(plus, r, t)
    assert [c.serialize() for c in dev] == [c.serialize() for c in a.coverages.values()]
    s.close(); t.close()


def test_json_dump_layout_is_the_references_byte_for_byte():
    """kvarq/util.py:272-294 writes an analysis as JSON with two levels indented (two blanks each) and everything
    below on one line.  The expected text is derived by hand from that rule: the reference feeds the chunks of
    Python 2.7's ``json.JSONEncoder(indent=2)`` -- whose item separator is ", " also when it indents, so that a
    line which is followed by another item ends in ", " -- through two regular expressions: one strips the line
    breaks and indentation of more than four blanks (levels three and deeper collapse onto their parent's line:
    `"fastq": ["a_1.fastq", "a_2.fastq"]`), the other holds back the line break in front of a bracket that closes
    such a collapsed level.  Empty containers stay `[]`; the top level closes on a line of its own."""
    import collections
    import io
    from kvarq_amd.analyse import json_dump
    data = collections.OrderedDict([
        ('analyses', collections.OrderedDict([('MTBC/spoligo', ['SIT 53', 'octal 777777777760771'])])),
        ('info', collections.OrderedDict([('fastq', ['a_1.fastq', 'a_2.fastq']), ('scantime', 1.5),
                                          ('config', collections.OrderedDict([('maxerrors', 2), ('Amin', '.')]))])),
        ('coverages', [['spacer1', '2-2-2 1[T]'], ['spacer2', '0-0-0 ']]),
        ('hits', [[0, 54, -12, 3, 51]]),
        ('empty', []),
        ('n', 3),
    ])
    expected = (
        '{\n'
        '  "analyses": {\n'
        '    "MTBC/spoligo": ["SIT 53", "octal 777777777760771"]\n'
        '  }, \n'
        '  "info": {\n'
        '    "fastq": ["a_1.fastq", "a_2.fastq"], \n'
        '    "scantime": 1.5, \n'
        '    "config": {"maxerrors": 2, "Amin": "."}\n'
        '  }, \n'
        '  "coverages": [\n'
        '    ["spacer1", "2-2-2 1[T]"], \n'
        '    ["spacer2", "0-0-0 "]\n'
        '  ], \n'
        '  "hits": [\n'
        '    [0, 54, -12, 3, 51]\n'
        '  ], \n'
        '  "empty": [], \n'
        '  "n": 3\n'
        '}')
    fd = io.StringIO()
    json_dump(data, fd)
    assert fd.getvalue() == expected
    import json
    assert json.loads(fd.getvalue(), object_pairs_hook=collections.OrderedDict) == json.loads(json.dumps(data), object_pairs_hook=collections.OrderedDict)
