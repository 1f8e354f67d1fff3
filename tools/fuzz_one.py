#!/usr/bin/env python3
"""One case of tools/fuzz_parity.py in detail: what differs between the GPU engine and the oracle.

usage: python tools/fuzz_one.py <seed> [seeded] [stride]
"""
import os, sys, random, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_parity as F
from kvarq_amd import engine
from oracle import oracle as O

seed = int(sys.argv[1]); seeded = len(sys.argv) > 2 and sys.argv[2] == 'seeded'
data, seqs, cfg = F.make_case(seed, seeded)
if seeded:
    os.environ['KVQ_STRIDE'] = sys.argv[3] if len(sys.argv) > 3 else str(random.Random(seed).choice([2, 4, 8, 8]))
print('seed', seed, 'stride', os.environ.get('KVQ_STRIDE'), 'bytes', len(data), 'cfg', cfg, 'seq lengths', [len(s) for s in seqs])
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, 'c.fastq'); open(p, 'wb').write(data)
    engine.config(**cfg)
    g = F.outcome(lambda: engine.findseqs(p, seqs)); o = F.outcome(lambda: O.findseqs(p, seqs, **cfg))
print('kinds', g[0], o[0])
if g[0] == 'ok' and o[0] == 'ok':
    gs, os_ = set(g[1]), set(o[1])
    print('hits gpu %d oracle %d; only gpu %d, only oracle %d' % (len(g[1]), len(o[1]), len(gs - os_), len(os_ - gs)))
    for h in sorted(gs - os_)[:10]: print('  only gpu   ', h)
    for h in sorted(os_ - gs)[:10]: print('  only oracle', h)
    if gs == os_ and g[1] != o[1]:
        for i, (a, b) in enumerate(zip(g[1], o[1])):
            if a != b: print('  first order difference at', i, a, b); break
    print('stats equal', g[3] == o[3])
    for k in g[3]:
        if g[3][k] != o[3][k]:
            if isinstance(g[3][k], tuple):
                d = [(i, a, b) for i, (a, b) in enumerate(zip(g[3][k], o[3][k])) if a != b]
                print('   ', k, 'len', len(g[3][k]), len(o[3][k]), 'differs at (index, gpu, oracle)', d[:6])
            else:
                print('   ', k, g[3][k], o[3][k])
    # the records of the differing hits
    lines = data.split(b'\n')
    for h in sorted((gs ^ os_))[:4]:
        at = h.file_pos; rec = data[at:at + 700].split(b'\n')[:4]
        print('  record at', at, [len(x) for x in rec], 'seq', h.seq_nr, 'len', len(seqs[h.seq_nr]))
else:
    print(g[1] if g[0] != 'ok' else '', o[1] if o[0] != 'ok' else '')
