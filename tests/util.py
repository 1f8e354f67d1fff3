"""helpers shared by the oracle and HIP parity tests"""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

_expected = None


def expected():
    global _expected
    if _expected is None:
        with open(os.path.join(HERE, 'golden', 'expected.json')) as f:
            _expected = json.load(f)
    return _expected


def canon(result, order):
    """hits + hitseqs of an engine-shaped result as JSON-able lists, in stored order"""
    hits = [list(h) for h in result['hits']]
    hitseqs = [h.decode('latin-1') if isinstance(h, bytes) else h for h in result['hitseqs']]
    if order == 'sorted':
        pairs = sorted(zip(hits, hitseqs))
        hits, hitseqs = [p[0] for p in pairs], [p[1] for p in pairs]
    return hits, hitseqs


def check_against_expected(name, result, exp=None):
    """assert that an engine-shaped result equals the stored reference outcome"""
    exp = exp or expected()[name]
    hits, hitseqs = canon(result, exp['order'])
    assert len(hits) == exp['n_hits'], (name, len(hits), exp['n_hits'])
    if 'hits' in exp:
        assert hits == exp['hits'], name
        assert hitseqs == exp['hitseqs'], name
    assert hashlib.sha256(json.dumps([hits, hitseqs]).encode()).hexdigest() == exp['hits_sha256'], name
    st, est = result['stats'], exp['stats']
    # the reference reads past its 1024-entry histogram when a read is longer
    # (workhorse.c:1214-1216): only the in-bounds part and the length are defined
    assert len(st['readlengths']) == len(est['readlengths']), name
    assert list(st['readlengths'])[:1024] == est['readlengths'][:1024], name
    assert all(v == 0 for v in list(st['readlengths'])[1024:]), name
    for k in ('nseqhits', 'nseqbasehits'):
        assert list(st[k]) == est[k], (name, k)
    for k in ('parsed', 'total', 'records_parsed', 'sigints'):
        assert st[k] == est[k], (name, k, st[k], est[k])
    assert abs(st['progress'] - est['progress']) < 1e-6, (name, st['progress'], est['progress'])
