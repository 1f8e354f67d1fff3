"""
The oracle restatement (oracle/kvarq_oracle.c) against the stored outcomes of
the reference C engine (tests/golden/expected.json, written by
tests/golden/make_golden.py from oracle/_ref) on every case of tests/cases.py.
"""
import pytest

import cases
from oracle import oracle as O
from util import expected, check_against_expected

CASES = cases.all_cases(big=True)


@pytest.mark.parametrize('case', [c for c in CASES if c.ref_ok], ids=lambda c: c.name)
def test_oracle_matches_reference_outcome(case, tmp_path):
    exp = expected()[case.name]
    assert case.input_digest() == exp['input_sha256'], 'generated input drifted from the one the golden was made on'
    files = case.materialize(tmp_path)
    fn = files[0] if len(files) == 1 else files
    if 'error' in exp:
        kind, msg = exp['error']
        etype = {'format': O.OracleFormatError, 'OSError': IOError, 'IOError': IOError,
                 'RuntimeError': RuntimeError, 'MemoryError': MemoryError}[kind]
        with pytest.raises(etype) as ei:
            O.findseqs(fn, case.seq_bytes(), **case.config)
        assert str(ei.value) == msg
        return
    r = O.findseqs(fn, case.seq_bytes(), **case.config)
    check_against_expected(case.name, r, exp)


def test_thread_count_does_not_change_the_result(tmp_path):
    """the restatement returns canonical order for any worker count"""
    case = cases.by_name()['multichunk']
    files = case.materialize(tmp_path)
    cfg = dict(case.config)
    cfg['nthreads'] = 1
    a = O.findseqs(files[0], case.seq_bytes(), **cfg)
    cfg['nthreads'] = 5
    b = O.findseqs(files[0], case.seq_bytes(), **cfg)
    assert a == b
