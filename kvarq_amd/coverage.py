"""
``kvarq.analyse.Coverage`` for Python 3 (reference kvarq/analyse.py:25-185), the
consumer of the scan: hits folded onto a template, per-position depth and the
multiset of non-matching bases.  Besides the reference's per-hit ``apply_hit``
it can be filled directly from the counter arrays the GPU fold produces
(``from_counters``), which removes the per-hit-base Python loop
(kvarq/analyse.py:379-381) from the path.
"""
from collections import OrderedDict

PAIRS = {'A': 'T', 'T': 'A', 'G': 'C', 'C': 'G', 'N': 'N'}      # kvarq/genes.py:204
MUT_LETTERS = 'ACGTN'                                            # counter order of the device fold (+ 'other')


class Sequence(object):
    """the part of kvarq.genes.Sequence the coverage needs (kvarq/genes.py:224-276)"""

    def __init__(self, bases, left=0, right=0, pos=None, plus_strand=True):
        self.bases, self.left, self.right, self.pos, self.plus_strand = bases, left, right, pos, plus_strand

    def __len__(self):
        return len(self.bases)

    def __getitem__(self, idx):
        return self.bases[idx]

    def reverse(self):
        return Sequence(''.join(PAIRS[b] for b in self.bases)[::-1], pos=self.pos,
                        plus_strand=not self.plus_strand, left=self.left, right=self.right)

    def plus_idx(self, idx):
        return idx if self.plus_strand else len(self.bases) - idx - 1

    def plus_base(self, base):
        return base if self.plus_strand else PAIRS[base]


class Coverage(object):

    def __init__(self, plus_seq):
        self.plus_seq = plus_seq
        self.minus_seq = plus_seq.reverse()
        self.coverage = [0] * len(plus_seq)
        self.mutations = {}
        self.start = plus_seq.left
        self.stop = len(plus_seq) - plus_seq.right

    def apply_hit(self, hit, hitseq, on_plus_strand):
        """kvarq/analyse.py:57-78"""
        seq = self.plus_seq if on_plus_strand else self.minus_seq
        start = max(0, hit.seq_pos)
        for i, j in enumerate(range(start, start + hit.length)):
            c_j = seq.plus_idx(j)
            c_b = seq.plus_base(hitseq[i])
            self.coverage[c_j] += 1
            if hitseq[i] != seq[j]:
                self.mutations[c_j] = self.mutations.get(c_j, '') + c_b

    def add_counters(self, cov, mut, on_plus_strand):
        """add the device fold of ONE sequence (cov[len], mut[len*6]: A,C,G,T,N,other per
        position, indexed along that sequence) -- the same update apply_hit makes hit by hit"""
        n = len(self.coverage)
        for j in range(n):
            c_j = j if on_plus_strand else n - j - 1
            self.coverage[c_j] += int(cov[j])
            m = mut[6 * j:6 * j + 6]
            if int(m[5]):
                raise KeyError('read base outside ACGTN on a covered position')   # Sequence.pairs has no such key
            for k, letter in enumerate(MUT_LETTERS):
                if int(m[k]):
                    b = letter if on_plus_strand else PAIRS[letter]
                    self.mutations[c_j] = self.mutations.get(c_j, '') + b * int(m[k])

    def bases_at(self, idx):
        m = self.mutations.get(idx, '')
        ret = {self.plus_seq[idx]: self.coverage[idx] - len(m)}
        for b in set(m):
            ret[b] = m.count(b)
        return ret

    def fractions_at(self, idx):
        bases = self.bases_at(idx)
        total = sum(bases.values())
        return OrderedDict(sorted([(b, n / float(max(1, total))) for b, n in bases.items()], key=lambda x: -x[1]))

    def minf(self, include_margins=False):
        start, stop = (0, len(self)) if include_margins else (self.start, self.stop)
        return min(list(self.fractions_at(pos).values())[0] for pos in range(start, stop))

    def mixed(self, fmin=0.9, include_margins=False):
        cminf = self.minf(include_margins=include_margins)
        return cminf > 0 and cminf < fmin

    def mean(self, include_margins=True):
        if include_margins:
            return sum(self.coverage) / float(len(self.coverage))
        return sum(self.coverage[self.start:self.stop]) / float(self.stop - self.start)

    def serialize(self):
        """kvarq/analyse.py:157-164"""
        cov = '-'.join(str(c) for c in self.coverage)
        mut = '-'.join('%d[%s]' % (idx, ''.join(sorted(self.mutations[idx]))) for idx in sorted(self.mutations))
        return cov + ' ' + mut

    def deserialize(self, text):
        c_s, _, m_s = text.partition(' ')
        self.coverage = [int(x) for x in c_s.split('-')]
        self.mutations = dict((int(x[:x.index('[')]), x[x.index('[') + 1:x.index(']')]) for x in m_s.split('-')) if m_s else {}

    def __len__(self):
        return len(self.coverage)

    def __getitem__(self, idx):
        return self.coverage[idx]


def coverages_from_scan(plus_seqs, result, table):
    """one Coverage per template from a Scanner.finish() result scanned with
    ``plus + reverse complements`` (kvarq/analyse.py:352-354, 379-381)"""
    n = len(plus_seqs)
    out = []
    for k, bases in enumerate(plus_seqs):
        text = bases.decode('latin-1') if isinstance(bases, bytes) else bases
        c = Coverage(Sequence(text))
        for s, plus in ((k, True), (k + n, False)):
            a, b = table.seq_offset[s], table.seq_offset[s + 1]
            c.add_counters(result['coverage'][a:b], result['mutations'][6 * a:6 * b], plus)
        out.append(c)
    return out
