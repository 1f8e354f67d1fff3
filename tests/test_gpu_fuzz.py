"""A fixed slice of the randomised parity campaign (tools/fuzz_parity.py): generated FastQ files
(ragged reads, N and stray bytes, headers and '+' lines with text, CR LF, truncated tails, malformed
records), tables cut from them and random configurations; engine.findseqs on the GPU must equal the
oracle bit for bit -- hits, hit bytes, every statistic, and the message of a format error."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import fuzz_parity as F                                        # noqa: E402
from kvarq_amd import engine                                   # noqa: E402
from oracle import oracle as O                                 # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seeded,first,count', [(False, 1, 60), (True, 2000, 90)])
def test_generated_cases_match_the_oracle(seeded, first, count, tmp_path, monkeypatch):
    hits = formats = 0
    for seed in range(first, first + count):
        data, seqs, cfg = F.make_case(seed, seeded)
        if not data:
            continue
        p = str(tmp_path / ('c%d.fastq' % seed))
        with open(p, 'wb') as f:
            f.write(data)
        if seeded:
            monkeypatch.setenv('KVQ_STRIDE', str(random.Random(seed).choice([2, 4, 8, 8])))
        engine.config(**cfg)
        g = F.outcome(lambda: engine.findseqs(p, seqs))
        o = F.outcome(lambda: O.findseqs(p, seqs, **cfg))
        assert g == o, 'seed %d (%s), cfg %r' % (seed, 'seeded' if seeded else 'general', cfg)
        hits += len(g[1]) if g[0] == 'ok' else 0
        formats += g[0] == 'format'
        os.unlink(p)
    assert hits > 1000 and formats >= 1          # the slice does exercise both outcomes
