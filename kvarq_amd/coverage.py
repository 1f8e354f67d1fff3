"""
Coverage of a template, kept the way the GPU fold produces it: counts.

The scan (``kvq_fold_batch`` / ``kvq_cov_apply``) leaves, per table position, a depth and six counters of
non-matching read bases (A, C, G, T, N, other).  ``Coverage`` holds exactly that for one template, in
plus-strand coordinates -- ``depth[len]`` and ``alt[len, 5]`` as numpy arrays -- and answers the questions
the reference's ``kvarq.analyse.Coverage`` answers (kvarq/analyse.py:25-185: depth per position, bases and
their fractions at a position, ``minf``, ``mixed``, ``mean``, the ``.json`` string) from the counts, vectorised.
It is filled either straight from the counter arrays of a scan (``add_counters``: no per-hit work at all) or
hit by hit (``apply_hit``, the reference's update, kvarq/analyse.py:57-78 -- here one numpy statement per hit
instead of a Python loop per base).

Only the multiset of non-matching bases at a position matters downstream (the reference sorts it when it
writes it and counts it when it reads it, kvarq/analyse.py:80-87, 162), which is what makes counts enough.
The one thing counts cannot hold is a non-matching read byte outside ACGTN on the plus strand (the
reference keeps the byte itself; on the minus strand it fails with KeyError): those are kept, sparsely, in
``other``.
"""
from collections import OrderedDict

import numpy as np

LETTERS = 'ACGTN'                                       # column order of `alt` = counter order of the device fold
_COL = np.full(256, 5, dtype=np.int64)                  # byte -> column (5 = outside ACGTN)
for _i, _c in enumerate(LETTERS):
    _COL[ord(_c)] = _i
_COMPLEMENT_COL = np.array([3, 2, 1, 0, 4], dtype=np.int64)          # A<->T, C<->G, N<->N (kvarq/genes.py:204)
_COMPLEMENT = bytes.maketrans(b'ACGTN', b'TGCAN')
_SORTED_COLS = sorted(range(5), key=lambda k: LETTERS[k])             # columns in the order sorted() puts the letters: A C G N T


class Sequence(object):
    """a template's bases with its margins; the minus strand is the reverse complement, and positions /
    bases seen on it map back to the plus strand (what kvarq.genes.Sequence gives the coverage,
    kvarq/genes.py:224-276)"""

    def __init__(self, bases, left=0, right=0, pos=None, plus_strand=True):
        self.bases, self.left, self.right, self.pos, self.plus_strand = bases, left, right, pos, plus_strand

    def __len__(self):
        return len(self.bases)

    def __getitem__(self, idx):
        return self.bases[idx]

    def reverse(self):
        # (a base outside ACGTN has no complement: KeyError, as in the reference)
        for b in self.bases:
            if b not in LETTERS:
                raise KeyError(b)
        flipped = self.bases.encode('latin-1').translate(_COMPLEMENT)[::-1].decode('latin-1')
        return Sequence(flipped, left=self.left, right=self.right, pos=self.pos, plus_strand=not self.plus_strand)

    def plus_idx(self, idx):
        return idx if self.plus_strand else len(self.bases) - 1 - idx

    def plus_base(self, base):
        if self.plus_strand:
            return base
        return LETTERS[_COMPLEMENT_COL[LETTERS.index(base)]] if base in LETTERS else {}[base]      # KeyError outside ACGTN


class Coverage(object):

    def __init__(self, plus_seq):
        self.plus_seq = plus_seq
        self.minus_seq = plus_seq.reverse()
        n = len(plus_seq)
        self.depth = np.zeros(n, dtype=np.int64)                # reads over each position
        self.alt = np.zeros((n, 5), dtype=np.int64)             # of those: read bases that differ from the template, by letter (plus strand)
        self.other = {}                                         # position -> differing read bytes outside ACGTN (plus strand only), as a string
        self.start = plus_seq.left
        self.stop = n - plus_seq.right
        self._ref = np.frombuffer(plus_seq.bases.encode('latin-1'), dtype=np.uint8)
        self._ref_minus = np.frombuffer(self.minus_seq.bases.encode('latin-1'), dtype=np.uint8)

    # ---- filling -------------------------------------------------------------------------------------------

    def clear(self):
        self.depth[:] = 0
        self.alt[:] = 0
        self.other = {}

    def apply_hit(self, hit, hitseq, on_plus_strand):
        """one hit of the scan (kvarq/analyse.py:57-78): depth + 1 over the hit, the read's bases where they
        differ from the strand it was found on"""
        n = len(self.depth)
        start = max(0, hit.seq_pos)
        j = np.arange(start, start + hit.length)
        read = np.frombuffer((hitseq if isinstance(hitseq, bytes) else hitseq.encode('latin-1'))[:hit.length], dtype=np.uint8)
        where = j if on_plus_strand else n - 1 - j              # plus-strand positions
        np.add.at(self.depth, where, 1)
        differs = read != (self._ref if on_plus_strand else self._ref_minus)[j]
        if not differs.any():
            return
        col = _COL[read[differs]]
        at = where[differs]
        odd = col == 5
        if odd.any():
            if not on_plus_strand:
                raise KeyError(chr(read[differs][odd][0]))      # the reference: Sequence.pairs has no such key
            for p, byte in zip(at[odd].tolist(), read[differs][odd].tolist()):
                self.other[p] = self.other.get(p, '') + chr(byte)
        col, at = col[~odd], at[~odd]
        np.add.at(self.alt, (at, col if on_plus_strand else _COMPLEMENT_COL[col]), 1)

    def add_counters(self, cov, mut, on_plus_strand):
        """the device fold of ONE sequence of the scan's table (cov[len], mut[len * 6]: A, C, G, T, N, other per
        position, indexed along that sequence) -- the sum of what apply_hit does for each of its hits"""
        n = len(self.depth)
        cov = np.asarray(cov, dtype=np.int64)
        m = np.asarray(mut, dtype=np.int64).reshape(n, 6)
        if m[:, 5].any():
            raise KeyError('read base outside ACGTN on a covered position')     # (which byte it was is not in the counters)
        if on_plus_strand:
            self.depth += cov
            self.alt += m[:, :5]
        else:
            self.depth += cov[::-1]
            self.alt += m[::-1, :5][:, _COMPLEMENT_COL]

    # ---- what the callers ask ------------------------------------------------------------------------------

    @property
    def coverage(self):
        """depth per position as a list (the reference's attribute)"""
        return self.depth.tolist()

    @property
    def mutations(self):
        """position -> the differing read bases there, sorted (the reference's attribute, as it writes it)"""
        out = {}
        for p in np.flatnonzero(self.alt.any(axis=1)).tolist() + [q for q in self.other if not self.alt[q].any()]:
            out[p] = ''.join(sorted(''.join(LETTERS[k] * int(self.alt[p, k]) for k in range(5)) + self.other.get(p, '')))
        return out

    def bases_at(self, idx):
        """base -> number of reads that show it at position idx"""
        ret = {self.plus_seq[idx]: int(self.depth[idx] - self.alt[idx].sum()) - len(self.other.get(idx, ''))}
        for k in range(5):
            if self.alt[idx, k]:
                ret[LETTERS[k]] = int(self.alt[idx, k])
        for b in set(self.other.get(idx, '')):
            ret[b] = self.other[idx].count(b)
        return ret

    def fractions_at(self, idx):
        bases = self.bases_at(idx)
        total = max(1, sum(bases.values()))
        return OrderedDict(sorted(((b, c / float(total)) for b, c in bases.items()), key=lambda x: -x[1]))

    def _top_fraction(self):
        """per position: the share of the most frequent base"""
        others = np.zeros(len(self.depth), dtype=np.int64)
        top_other = np.zeros(len(self.depth), dtype=np.int64)
        for p, s in self.other.items():
            others[p] = len(s)
            top_other[p] = max(s.count(b) for b in set(s))
        ref = self.depth - self.alt.sum(axis=1) - others
        top = np.maximum(np.maximum(ref, self.alt.max(axis=1)), top_other)
        return top / np.maximum(1, self.depth).astype(np.float64)

    def minf(self, include_margins=False):
        lo, hi = (0, len(self.depth)) if include_margins else (self.start, self.stop)
        return float(self._top_fraction()[lo:hi].min())

    def mixed(self, fmin=0.9, include_margins=False):
        f = self.minf(include_margins=include_margins)
        return 0 < f < fmin

    def mean(self, include_margins=True):
        d = self.depth if include_margins else self.depth[self.start:self.stop]
        return float(d.sum()) / len(d)

    # ---- the .json string (kvarq/analyse.py:157-185): depths joined by '-', a blank, "position[bases]" joined by '-' ----

    def serialize(self):
        mut = self.mutations
        return '-'.join(map(str, self.depth.tolist())) + ' ' + '-'.join('%d[%s]' % (p, mut[p]) for p in sorted(mut))

    def deserialize(self, text):
        depths, _, muts = text.partition(' ')
        self.depth = np.array([int(x) for x in depths.split('-')], dtype=np.int64)
        self.alt = np.zeros((len(self.depth), 5), dtype=np.int64)
        self.other = {}
        for item in (muts.split('-') if muts else ()):
            p, letters = int(item[:item.index('[')]), item[item.index('[') + 1:item.index(']')]
            for b in letters:
                if b in LETTERS:
                    self.alt[p, LETTERS.index(b)] += 1
                else:
                    self.other[p] = self.other.get(p, '') + b

    def __len__(self):
        return len(self.depth)

    def __getitem__(self, idx):
        return int(self.depth[idx])


def coverages_from_scan(plus_seqs, result, table):
    """one Coverage per template from a Scanner.finish() result scanned with ``plus + reverse complements``
    (kvarq/analyse.py:352-354, 379-381): the counter arrays go in as they are"""
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
