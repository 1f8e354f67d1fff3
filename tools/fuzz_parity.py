#!/usr/bin/env python3
"""Randomised parity campaign: engine.findseqs on the GPU against the oracle (and, where it loads and
the case is safe for it, the reference engine) on generated FastQ files.

usage: python tools/fuzz_parity.py [first seed] [number of cases] [seeded]

Every case draws a file (ragged read lengths, N and stray bytes among the bases, headers and '+'
lines with text, CR LF, truncated tails, empty lines, now and then a malformed record), a sequence
table cut from its reads (exact, mutated, reverse-complemented, short and long, sometimes non-ACGT)
and an engine configuration; hits, hit bytes and all statistics must agree bit for bit, and a
malformed file must raise the same message.  Prints one line per failing seed; exit status 1 if any.
"""
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from kvarq_amd import engine                                   # noqa: E402
from kvarq_amd.fastq import FastqFileFormatException           # noqa: E402
from oracle import oracle as O                                 # noqa: E402

COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N'}


def make_case(seed, seeded=False):
    rnd = random.Random(seed)
    e_seed = rnd.choice([0, 1, 2, 2, 2, 3]); need = 8 * (e_seed + 1); lo = need + rnd.choice([1, 1, 3, 7, 20])
    style = rnd.choice(['illumina', 'ragged', 'ragged', 'long', 'mixed']) if seeded else rnd.choice(['short', 'illumina', 'ragged', 'long', 'mixed'])
    nrec = rnd.choice([1, 2, 5, 40, 300, 1500, 4000]) if style != 'long' else rnd.choice([1, 3, 20])
    eol = '\r\n' if rnd.random() < 0.08 else '\n'
    qual_lo, qual_hi = rnd.choice([(33, 74), (35, 74), (45, 50), (64, 104), (40, 41)])
    pn = rnd.choice([0.0, 0.0, 0.01, 0.2])
    pbad = rnd.choice([0.0, 0.01, 0.01, 0.05, None])
    stray = rnd.random() < 0.1
    recs, reads = [], []
    # most cases sample their reads from a small genome (both strands, a few substitutions), so that a
    # sequence cut from it is met by many reads on many diagonals (all alignment classes)
    genome = ''.join(rnd.choice('ACGT') for _ in range(rnd.choice([400, 2000, 6000]))) if rnd.random() < (0.85 if seeded else 0.5) else None
    for i in range(nrec):
        if style == 'short':
            L = rnd.randint(0, 40)
        elif style == 'illumina':
            L = 150
        elif style == 'ragged':
            L = rnd.randint(20, 400)
        elif style == 'long':
            L = rnd.choice([600, 1023, 1024, 1500, 4200, 9000])
        else:
            L = rnd.choice([0, 1, 7, 8, 24, 25, 26, 51, 100, 150, 151, 300])
        if genome and L <= len(genome):
            st = rnd.randint(0, len(genome) - L)
            frag = genome[st:st + L]
            if rnd.random() < 0.5:
                frag = ''.join(COMP[c] for c in reversed(frag))
            bases = ''.join('N' if rnd.random() < pn else (rnd.choice('ACGT') if rnd.random() < 0.01 else c) for c in frag)
        else:
            bases = ''.join('N' if rnd.random() < pn else rnd.choice('ACGT') for _ in range(L))
        if stray and L and rnd.random() < 0.05:
            k = rnd.randrange(L)
            bases = bases[:k] + rnd.choice('acgtXn.-') + bases[k + 1:]
        Lq = L if rnd.random() > 0.02 else max(0, L + rnd.choice([-1, 1, 3]))
        if seeded and pbad is not None:
            quals = ''.join('#' if rnd.random() < pbad else rnd.choice('5?FI') for _ in range(Lq))       # mostly good scores, so that reads stay long
        else:
            quals = ''.join(chr(rnd.randint(qual_lo, qual_hi)) for _ in range(Lq))
        if Lq and rnd.random() < 0.05:
            quals = rnd.choice('+@') + quals[1:]                  # score lines may start with '+' or '@'
        head = '@r%d %s' % (i, ''.join(rnd.choice('abc:/ 0123456789#+-') for _ in range(rnd.randint(0, 30))))
        plus = '+' if rnd.random() < 0.8 else '+' + head[1:]
        recs.append(head + eol + bases + eol + plus + eol + quals + eol)
        reads.append(bases)
    text = ''.join(recs)
    r = rnd.random()
    if r < 0.05:
        text = text[:max(0, len(text) - rnd.randint(1, 60))]              # truncated tail
    elif r < 0.10:
        text += eol * rnd.randint(1, 3)                                    # empty lines behind the last record
    elif r < 0.14 and nrec > 2:                                            # a malformed record somewhere
        k = rnd.randrange(nrec)
        bad = recs[k]
        bad = ('X' + bad[1:]) if rnd.random() < 0.5 else bad.replace(eol + '+', eol + '-', 1)
        text = ''.join(recs[:k]) + bad + ''.join(recs[k + 1:])
    # sequences: cut from reads (so that something is found), mutated, reverse-complemented, or random
    nseq = rnd.choice([1, 2, 5, 12, 40])
    seqs = []
    src = ([genome] if genome else []) + [b for b in reads if len(b) >= 3] or ['ACGTACGTACGT']
    for _ in range(nseq):
        b = genome if genome and rnd.random() < 0.7 else rnd.choice(src)
        ln = min(len(b), max(lo if seeded else 0, rnd.choice([3, 8, 20, 24, 25, 26, 30, 51, 100, 150, 300, 611, 1200])))
        st = rnd.randint(0, len(b) - ln)
        s = list(b[st:st + ln])
        if rnd.random() < 0.3:                                             # hang over the end of the read
            s = s + [rnd.choice('ACGT') for _ in range(rnd.randint(1, 40))]
        if rnd.random() < 0.3:
            s = [rnd.choice('ACGT') for _ in range(rnd.randint(1, 40))] + s
        for _m in range(rnd.choice([0, 0, 1, 2, 3]) if not seeded else rnd.randint(0, e_seed + 1)):
            s[rnd.randrange(len(s))] = rnd.choice('ACGT')
        s = ''.join(s)
        if rnd.random() < 0.4:
            s = ''.join(COMP.get(c, 'N') for c in reversed(s))
        seqs.append(s.encode('latin-1'))
    if rnd.random() < 0.2:
        seqs.append(bytes(rnd.choice(b'ACGT') for _ in range(rnd.randint(1, 60))))
    cfg = dict(maxerrors=rnd.choice([0, 1, 2, 2, 3]), minoverlap=rnd.choice([1, 8, 16, 24, 25, 32, 40]),
               minreadlength=rnd.choice([1, 8, 24, 25, 40]), Amin=chr(rnd.choice([33, 35, 46, 46, 46, 60, 70])), Azero='!',
               nthreads=rnd.choice([1, 2, 4]))
    if seeded:
        # a configuration and a table the seed-filter kernel takes: (e+1)*8 <= minoverlap, minreadlength; ACGT only
        e = e_seed
        cfg.update(maxerrors=e, minoverlap=need + rnd.choice([0, 1, 5, 20]), minreadlength=need + rnd.choice([0, 1, 10]),
                   Amin=chr(rnd.choice([35, 46, 46, 60])))
        seqs = [s for s in seqs if len(s) >= lo and set(s) <= set(b'ACGT')] or [(b'ACGTTGCA' * 20)[:lo + 3]]
    return text.encode('latin-1'), seqs, cfg


def outcome(fn):
    try:
        r = fn()
        return ('ok', tuple(r['hits']), [bytes(h) for h in r['hitseqs']], r['stats'])
    except (FastqFileFormatException, O.OracleFormatError) as e:
        return ('format', str(e))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    seeded = len(sys.argv) > 3 and sys.argv[3] == 'seeded'      # bias towards what the seed-filter kernel takes; random KVQ_STRIDE
    bad = 0
    kinds = {'ok': 0, 'format': 0}
    nhits = 0
    with tempfile.TemporaryDirectory() as d:
        for seed in range(first, first + count):
            data, seqs, cfg = make_case(seed, seeded)
            if seeded:
                os.environ['KVQ_STRIDE'] = str(random.Random(seed).choice([2, 4, 8, 8]))
            p = os.path.join(d, 'c%d.fastq' % seed)
            with open(p, 'wb') as f:
                f.write(data)
            if not data:
                continue
            engine.config(**cfg)
            g = outcome(lambda: engine.findseqs(p, seqs))
            o = outcome(lambda: O.findseqs(p, seqs, **cfg))
            kinds[g[0]] = kinds.get(g[0], 0) + 1
            if g[0] == 'ok':
                nhits += len(g[1])
            if g != o:
                bad += 1
                what = 'kind' if g[0] != o[0] else ('message' if g[0] == 'format' else
                        'hits' if g[1] != o[1] else 'hitseqs' if g[2] != o[2] else 'stats')
                print('MISMATCH seed=%d (%s): %d bytes, %d sequences, cfg=%r' % (seed, what, len(data), len(seqs), cfg))
                sys.stdout.flush()
            os.unlink(p)
    print('%d cases from seed %d: %d mismatches (%r, %d hits in all)' % (count, first, bad, kinds, nhits))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
