#!/usr/bin/env python3
"""
tests/golden/make_golden.py -- regenerate tests/golden/expected.json.

Runs every case of tests/cases.py through the REFERENCE C engine (the
unmodified /root/reference/csrc/workhorse.c compiled into oracle/_ref by
oracle/build_ref.sh) and stores what it returned: hits, hitseqs, stats, or the
exception it raised.  Only runs where /root/reference exists (this container);
the JSON it writes is the committed fixture that travels to the GPU box.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import cases                      # noqa: E402
from oracle import oracle as O    # noqa: E402


def main():
    O.build()
    eng = O.ref_engine()
    if eng is None:
        raise SystemExit('oracle/_ref could not be built or loaded')
    sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    from kvarq.fastq import FastqFileFormatException
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for c in cases.all_cases(big=True):
            if not c.ref_ok:
                continue
            files = c.materialize(tmp)
            cfg = dict(c.config)
            entry = {'note': c.note, 'config': cfg, 'input_sha256': c.input_digest(), 'nseq': len(c.seqs)}
            try:
                r = O.ref_findseqs(files[0] if len(files) == 1 else files, c.seq_bytes(), **cfg)
            except FastqFileFormatException as e:
                entry['error'] = ['format', str(e)]
            except (IOError, RuntimeError, MemoryError) as e:
                entry['error'] = [type(e).__name__, str(e)]
            else:
                hits = [list(h) for h in r['hits']]
                hitseqs = [h.decode('latin-1') for h in r['hitseqs']]
                if cfg['nthreads'] > 1:
                    # worker interleaving is not deterministic in the reference:
                    # store the canonical (sorted) multiset, SURVEY 8a-1
                    pairs = sorted(zip(hits, hitseqs))
                    hits, hitseqs = [p[0] for p in pairs], [p[1] for p in pairs]
                    entry['order'] = 'sorted'
                else:
                    entry['order'] = 'scan'
                st = dict(r['stats'])
                st['readlengths'] = list(st['readlengths'])
                st['nseqhits'] = list(st['nseqhits'])
                st['nseqbasehits'] = list(st['nseqbasehits'])
                entry['n_hits'] = len(hits)
                if len(hits) <= 3000:
                    entry['hits'] = hits
                    entry['hitseqs'] = hitseqs
                entry['hits_sha256'] = hashlib.sha256(json.dumps([hits, hitseqs]).encode()).hexdigest()
                entry['stats'] = st
            out[c.name] = entry
            print('%-28s %s' % (c.name, entry.get('error') or '%d hits, %d records' % (entry['n_hits'], entry['stats']['records_parsed'])), flush=True)
    with open(os.path.join(HERE, 'expected.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('wrote', os.path.join(HERE, 'expected.json'))


if __name__ == '__main__':
    main()
