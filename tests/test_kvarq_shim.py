"""The reference's import names: ``from kvarq import engine`` (the drop-in boundary; setup.py:31-35,
csrc/workhorse.c:1567-1609) bound to the MI355X engine by the repository's ``kvarq/`` shim package.  The GPU tests
restate assertions of the reference's own tests/test_engine.py and tests/test_analyser.py on copies of its fixture
files, through that import name."""
import os

import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def kvarq_engine():
    """`from kvarq import engine` as a caller of the reference writes it; the counting SIGINT handler the import installs
    (as the C extension does) is taken out again behind the test"""
    import kvarq_amd.engine
    from kvarq import engine
    saved = engine.get_config()
    yield engine
    engine.config(**saved)
    kvarq_amd.engine.remove_sigint_counter()


def test_the_reference_import_names_resolve_to_this_engine(kvarq_engine):
    import kvarq
    import kvarq.fastq
    import kvarq.log
    import kvarq_amd.engine
    import kvarq_amd.fastq
    import logging
    engine = kvarq_engine
    assert os.path.dirname(os.path.abspath(kvarq.__file__)) == os.path.join(ROOT, 'kvarq')       # (not the oracle's stub package)
    # module contents: workhorse.c:1567-1596
    for name in ('config', 'get_config', 'findseqs', 'stats', 'stop', 'test', 'Hit'):
        assert getattr(engine, name) is getattr(kvarq_amd.engine, name)
    assert engine.Hit._fields == ('seq_nr', 'file_pos', 'seq_pos', 'length', 'readlength')
    # collaborators the C extension resolves at import: workhorse.c:1598-1609
    assert kvarq.fastq.FastqFileFormatException is kvarq_amd.fastq.FastqFileFormatException
    assert kvarq.log.lo is logging.getLogger('kvarq')
    assert set(engine.get_config()) == {'maxerrors', 'minoverlap', 'minreadlength', 'nthreads', 'Amin', 'Azero'}
    assert engine.test() is None


@pytest.mark.gpu
def test_engine_kats_through_the_reference_import_name(kvarq_engine, tmp_path):
    engine = kvarq_engine
    f = os.path.join(cases.FASTQS, 'test_engine.fastq')
    # test_engine.py:137-175 -- hits per sequence of the seven literal sequences
    engine.config(nthreads=1, maxerrors=0, minoverlap=1000, minreadlength=3, Amin='!', Azero='!')
    r = engine.findseqs(f, cases.SEQS_FINDSEQS)
    assert list(r['stats']['nseqhits']) == [19, 1, 0, 1, 1, 1, 1]
    assert len(r['hits']) == len(r['hitseqs']) == 24 and isinstance(r['hits'][0], engine.Hit)
    assert r['stats']['records_parsed'] == 14 and r['stats']['readlengths'][51] == 14
    # test_engine.py:178-205 -- gz == plain, paired == single (whole result)
    assert engine.findseqs(f + '.gz', cases.SEQS_FINDSEQS)['hits'] == r['hits']
    pair = [os.path.join(cases.FASTQS, 'test_engine_%d.fastq' % i) for i in (1, 2)]
    rp = engine.findseqs(pair, cases.SEQS_FINDSEQS)
    assert rp['hits'] == r['hits'] and rp['hitseqs'] == r['hitseqs'] and rp['stats'] == r['stats']
    # test_engine.py:208-224 -- 0, 1, 2, 3 hits at maxerrors 0 .. 3 (the last one on 6-base seeds)
    engine.config(minreadlength=25, minoverlap=25)
    for e in range(4):
        engine.config(maxerrors=e)
        assert len(engine.findseqs(f, cases.SEQS_MAXERROR)['hits']) == e
    # test_engine.py:336-346 -- the format errors, with the reference's messages
    from kvarq.fastq import FastqFileFormatException
    engine.config(maxerrors=2)
    with pytest.raises(FastqFileFormatException) as ei:
        engine.findseqs(os.path.join(cases.FASTQS, 'L3_N1014_hits_500_BROKEN.fastq'), cases.SPOLIGO)
    assert str(ei.value) == "record must start with '@' (and not '.') fpos=18534"
    with pytest.raises(IOError):
        engine.findseqs(str(tmp_path / 'absent.fastq'), ['ACGT'])


def spoligo_code(present):
    """the 15-digit code of a set of spacer numbers: fourteen octal digits for spacers 0..41, three a digit (a leading 4 =
    spacer 0 present, 1 and 2 absent), and a binary digit for spacer 42 (testsuites/MTBC/spoligo.py:12-29)"""
    value = sum(2 ** (41 - n) for n in present if n != 42)
    return '%014o' % value + ('1' if 42 in present else '0')


@pytest.mark.gpu
def test_spoligo_octal_of_the_analyser_fixture(kvarq_engine):
    """The one end-to-end result of the reference that is reproducible without its genome (SURVEY 8c iii): test_analyser.fastq
    against the 43 spoligotyping spacers at minoverlap=10, maxerrors=1 (tests/test_analyser.py:53-66) -> spacers 0 and 42 are
    found -> '400000000000001'.  A spacer is found when its mean coverage, margins excluded, is at least 2
    (kvarq/genes.py:326-332); the scan runs on 5-base seeds (kvq_seed_k), not in the exhaustive kernels."""
    from kvarq.fastq import Fastq
    from kvarq_amd import analyse, scan, synth
    engine = kvarq_engine
    engine.config(nthreads=1, minoverlap=10, maxerrors=1, minreadlength=10, Amin='!')
    spacers = synth.SPOLIGO_SPACERS                            # the plus strands (cases.SPOLIGO = these + their reverse complements)
    templates = dict(('spoligo%d' % i, s) for i, s in enumerate(spacers))
    a = analyse.Analyser()
    a.scan(Fastq(os.path.join(cases.FASTQS, 'test_analyser.fastq'), quiet=True), templates)
    a.update_coverages()
    assert a.stats['records_parsed'] == 72 and len(a.hits) == 4
    found = [i for i in range(43) if a['spoligo%d' % i].mean(include_margins=False) >= 2]
    assert found == [0, 42]
    assert spoligo_code(found) == '400000000000001'
    assert spoligo_code(range(43)) == '7' * 14 + '1' and spoligo_code([]) == '0' * 15 and spoligo_code([1, 2, 41]) == '300000000000010'
    t = scan.Table(cases.SPOLIGO, maxerrors=1, minoverlap=10, minreadlength=10, Amin='!')
    assert t.seed_k == 5 and all(t.seeded)
    t.close()
