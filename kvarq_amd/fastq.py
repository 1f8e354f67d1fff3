"""``kvarq.fastq`` for the MI355X engine: the exception the scan raises on malformed records, the PHRED
scale helpers callers derive ``Amin`` from, and ``Fastq`` -- the cheap look at a ``.fastq`` / ``.fastq.gz``
file that callers take before a scan (PHRED offset, read length, record estimate, the second file of a pair)
and after it (the record or the bases a hit came from).

The attribute and method names are the reference module's (kvarq/fastq.py:38-387: ``Analyser``, the CLI and the
tests reach for them); the implementation is this repository's own, built from three small parts:

* :class:`PhredScale` -- which vendor scales hold a range of score characters (pure arithmetic on a table);
* :func:`survey` -- ONE pass over a sample of records that validates each record and collects everything the
  constructor wants to know (score range, first read length, first record size);
* :class:`_Backtrack` -- finds the record around a file position from a block of bytes read backwards, in memory.

The scan itself never goes through this module (engine.findseqs reads the files in C++).
"""
import collections
import gzip
import itertools
import math
import os

from .log import lo


class FastqFileFormatException(Exception):
    """raised when a record does not start with '@' or its 3rd line not with '+' (csrc/workhorse.c:1037-1048
    resolves this class at import, 1598-1600), and by :class:`Fastq` for files it cannot make sense of"""


# --- PHRED scales ----------------------------------------------------------------------------------------

ASCII = ''.join(map(chr, range(33, 127)))          # the score characters, '!' first (kvarq/fastq.py:41-42)

VendorProperties = collections.namedtuple('VendorProperties', ['Qrange', 'dQ'])


class PhredScale(object):
    """the vendor scales: name -> (scores a file of that vendor may hold, offset of Q = 0 in ``ASCII``);
    data of kvarq/fastq.py:47-53, in its order (the order of ``Fastq.variants``)"""

    TABLE = collections.OrderedDict((
        ('Sanger', VendorProperties(range(0, 50), 0)),
        ('Solexa', VendorProperties(range(-5, 41), 31)),
        ('Illumina 1.3+', VendorProperties(range(0, 41), 31)),
        ('Illumina 1.5+', VendorProperties(range(3, 42), 31)),
        ('Illumina 1.8+', VendorProperties(range(0, 62), 0)),
    ))

    @classmethod
    def holding(cls, lowest, highest):
        """names of the scales whose score range holds both ASCII indices"""
        return [name for name, v in cls.TABLE.items() if lowest - v.dQ in v.Qrange and highest - v.dQ in v.Qrange]

    @classmethod
    def offset(cls, name):
        return cls.TABLE[name].dQ


VARIANTS = {name: v.dQ for name, v in PhredScale.TABLE.items()}


def Q2A(Q, variant='Sanger'):
    """ASCII character of PHRED score ``Q``; Q = 13 on Sanger / Illumina 1.8+ is '.', the product's default
    ``Amin`` (kvarq/fastq.py:245-247, kvarq/config.py:3)"""
    return ASCII[Q + PhredScale.offset(variant)]


# --- one pass over a sample of records -------------------------------------------------------------------

Record = collections.namedtuple('Record', ['identifier', 'bases', 'separator', 'scores'])
Survey = collections.namedtuple('Survey', ['records', 'lowest', 'highest', 'first'])

_BASES = frozenset('ACGTN')
_SCORES = frozenset(ASCII)


def check_record(rec):
    """the four format rules of a record (what kvarq/fastq.py:196-222 checks); raises FastqFileFormatException"""
    if not rec.identifier.startswith('@'):
        raise FastqFileFormatException('identifier (1st line of record) must begin with "@"')
    if not _BASES.issuperset(rec.bases):
        raise FastqFileFormatException('bases (2nd line of record) must contain only AGCTN')
    if rec.separator != '+' and rec.separator != '+' + rec.identifier[1:]:
        raise FastqFileFormatException('separator (3rd line of record) must be == "+" or "+(ident)"')
    extra = len(rec.scores) - len(rec.bases)
    if not (extra == 0 or (extra == 1 and rec.scores.endswith('!'))):
        raise FastqFileFormatException('bases must be ~ same length as phred score (2nd, 4th line)')
    if not _SCORES.issuperset(rec.scores):
        raise FastqFileFormatException('phred score (4th line of record) must contain only "%s"' % ASCII)


def _quota(n, points):
    """how many of ``n`` sample records each of ``points`` places of a file gets (the remainder goes to the last places)"""
    return [(k + 1) * n // points - k * n // points for k in range(points)]


def survey(fq, n, points, visit=None):
    """reads up to ``n`` records of ``fq`` -- spread over ``points`` places of a plain file, all from the start of a
    gzipped one (which cannot be entered anywhere else) -- checks each and returns a :class:`Survey`: how many were
    read, the lowest and highest score character seen (ASCII indices) and the first record.  ``visit(record)`` is
    called for every record.  An empty line ends the sample; behind it the file may hold empty lines only."""
    size = None if fq.gz else os.path.getsize(fq.fname)
    lowest, highest, count, first = 999, -999, 0, None
    fq.fd.seek(0)
    for place, want in enumerate(_quota(n, points)):
        if size is not None and place:
            fq.fd.seek(size * place // points)
            fq.seekback()
        for _ in range(want):
            lines = [fq._line() for _ in range(4)]
            if lines[0].rstrip('\r\n') == '':
                _nothing_but_empty_lines(fq, lines[1:])
                return Survey(count, lowest, highest, first)
            rec = Record(*(x.rstrip('\r\n') for x in lines))
            check_record(rec)
            if first is None:
                first = (rec, sum(map(len, lines)))
            if rec.scores:
                codes = [ord(c) - 33 for c in (min(rec.scores), max(rec.scores))]
                lowest, highest = min(lowest, codes[0]), max(highest, codes[1])
            if visit:
                visit(rec)
            count += 1
    return Survey(count, lowest, highest, first)


def _nothing_but_empty_lines(fq, already_read):
    for line in itertools.chain(already_read, iter(fq._line, '')):
        if line.rstrip('\r\n'):
            raise FastqFileFormatException('non-empty line after empty line (fpos=%d)' % fq.fd.tell())


# --- the record around a file position -------------------------------------------------------------------

class _Backtrack(object):
    """Line starts in front of a file position, nearest first, from blocks read backwards (one ``read`` per 64 KiB,
    not one per line)."""

    BLOCK = 1 << 16

    def __init__(self, fd, pos):
        self.fd = fd
        self.at = pos                # everything from here to the original position has been looked at

    def line_starts(self):
        """yields the start of the line that holds the position, then of the line in front of it, ..."""
        pending = self.at            # a line start is known to lie at or in front of this byte
        while pending > 0:
            lo_ = max(0, pending - self.BLOCK)
            self.fd.seek(lo_)
            block = self.fd.read(pending - lo_)
            cut = len(block)
            while True:
                nl = block.rfind(b'\n', 0, cut)
                if nl < 0:
                    break
                yield lo_ + nl + 1
                cut = nl             # (the newline itself ends the line in front)
            pending = lo_
        yield 0

    def first_byte(self, start):
        self.fd.seek(start)
        return self.fd.read(1)


def record_start(fd, pos):
    """start of the record whose '+' line is the nearest one at or in front of ``pos``: the position inside a record's
    separator or score line gives that record, inside its identifier or bases the one before (0 when there is none).
    A score line may begin with '+' as well; the separator is the '+' line whose line two above begins with '@'."""
    back = _Backtrack(fd, pos)
    window = collections.deque(maxlen=3)           # the last three line starts seen: [this, one below, two below]
    for start in back.line_starts():
        window.appendleft(start)
        # `start` is a candidate identifier line when the line two below it (seen two steps ago) begins with '+'
        if len(window) == 3 and back.first_byte(window[2]) == b'+' and back.first_byte(start) == b'@':
            return start
    return 0


# --- the class callers know --------------------------------------------------------------------------------

class Fastq(object):
    """``kvarq.fastq.Fastq``: a look at one FastQ file (or a ``_1`` / ``_2`` pair).

    After construction: ``fname``, ``fname2`` (or None), ``gz``, ``fd`` (binary file object), ``dQ`` / ``Azero`` /
    ``variants`` (the PHRED scale), ``readlength`` (of the first record), ``records_approx`` (None for ``.gz``).
    Text is latin-1 ``str`` (one character per byte)."""

    ASCII = ASCII
    vendor_variants = PhredScale.TABLE

    def __init__(self, fname, variant=None, fd=None, paired=False, quiet=False):
        self.fname, self.gz = fname, self._kind(fname)
        self.fd = fd or (gzip.GzipFile(fname, 'rb') if self.gz else open(fname, 'rb'))
        self.fname2 = self._mate(fname) if paired else None
        if self.fname2:
            lo.info('including paired file "%s"' % self.fname2)
        if not any(self.filesizes()):
            raise FastqFileFormatException('cannot scan empty file')
        seen = self._survey()                                        # (a malformed file is reported before a bad argument, as the reference does)
        if variant is not None and variant not in PhredScale.TABLE:
            raise FastqFileFormatException('unknown vendor variant "%s"' % variant)
        self.variants, self.dQ = self._scale(seen, variant)
        self.Azero = ASCII[self.dQ]
        rec, nbytes = seen.first if seen.first else (Record('', '', '', ''), 1)
        self.readlength = len(rec.bases)
        # records by the size of the first one; the mate of a pair is taken to hold as many
        self.records_approx = None if self.gz else os.path.getsize(fname) // max(1, nbytes) * len(self.filenames())
        if not quiet:
            lo.info('%sfastq : readlength=%s records_approx=%s dQ=%d variants=%s'
                    % ('gzipped ' if self.gz else '', '?' if self.gz else self.readlength, '?' if self.gz else self.records_approx, self.dQ, self.variants))

    # -- construction, piece by piece ------------------------------------------------------------------

    @staticmethod
    def _kind(fname):
        for ext, gz in (('.fastq.gz', True), ('.fastq', False)):
            if fname.endswith(ext):
                return gz
        raise FastqFileFormatException('fastq file must have extension ".fastq" or ".fastq.gz"')

    @staticmethod
    def _mate(fname):
        """``x_2.fastq[.gz]`` next to ``x_1.fastq[.gz]``, when it exists (the pairing rule of kvarq/fastq.py:90-98)"""
        stem, dot, ext = fname.rpartition('.fastq')
        if stem.endswith('_1'):
            mate = stem[:-2] + '_2' + dot + ext
            if os.path.exists(mate):
                return mate
        return None

    def _survey(self, n=1000, points=10, visit=None):
        if self.gz:
            lo.debug('gzipped fastq : scan %d points at start only' % n)
        return survey(self, n, points, visit)

    def _scale(self, seen, variant):
        """-> (names, dQ): the scales that hold the scores seen; a named ``variant`` overrides the guess"""
        fitting = PhredScale.holding(seen.lowest, seen.highest)
        lo.debug('min_pos=%d max_pos=%d' % (seen.lowest, seen.highest))
        if variant is not None:
            if variant not in fitting:
                lo.warning('specified vendor variant "%s" seems not to be compatible with file' % variant)
            return [variant], PhredScale.offset(variant)
        offsets = {PhredScale.offset(name) for name in fitting}
        if not offsets:
            raise FastqFileFormatException('could not find any suitable fastq vendor variant')
        if len(offsets) > 1:
            raise FastqFileFormatException('cannot determine dQ with guessed vendor variants "%s"' % fitting)
        return fitting, offsets.pop()

    # -- files -----------------------------------------------------------------------------------------

    def filenames(self):
        return [f for f in (self.fname, self.fname2) if f is not None]

    def filesizes(self):
        return [os.path.getsize(f) for f in self.filenames()]

    def _line(self):
        return self.fd.readline().decode('latin-1')

    # -- samples ---------------------------------------------------------------------------------------

    def min_max_score_check_file(self, n=1000, points=10):
        """format check of a sample of records; the smallest and largest score character seen, as indices into ``ASCII``"""
        seen = self._survey(n, points)
        return seen.lowest, seen.highest

    def lengths(self, Amin, n=1000, points=10):
        """quality-trimmed lengths (:meth:`cutoff`) of a sample of records"""
        out = []
        self._survey(n, points, visit=lambda rec: out.append(self.cutoff(rec.scores, Amin)[1]))
        return [x for x in out if x >= 0]

    @staticmethod
    def cutoff(scores, Amin):
        """``(start, length)`` of the longest run of scores >= Amin that is CLOSED by a worse score, the first of equally
        long ones; ``(0, -1)`` when no run is closed -- a run that reaches the end of the line does not count here
        (kvarq/fastq.py:295-308; the engine sees the closing newline and does count it, workhorse.c:1055-1068).  A line
        that begins with a bad score has the empty run (0, 0) closed by it."""
        best = (0, -1)
        at = 0
        runs = [(good, len(list(chars))) for good, chars in itertools.groupby(scores, key=lambda c: c >= Amin)]
        for k, (good, length) in enumerate(runs):
            closed = good and k + 1 < len(runs)
            if closed and length > best[1]:
                best = (at, length)
            if not good and k == 0 and best[1] < 0:
                best = (0, 0)
            at += length
        return best

    # -- PHRED arithmetic ------------------------------------------------------------------------------

    def A2Q(self, A):
        return ASCII.index(A) - self.dQ

    def Q2A(self, Q):
        return ASCII[Q + self.dQ]

    @staticmethod
    def Q2p(Q):
        return 10 ** (-.1 * Q)

    @staticmethod
    def p2Q(p):
        return int(-10 * math.log(p) / math.log(10))

    # -- records and hits by file position ---------------------------------------------------------------

    def seekback(self):
        """moves the file position to the start of the record it stands in (behind the record's '+' line) or of the one
        before (in front of it): see :func:`record_start`"""
        self.fd.seek(record_start(self.fd, self.fd.tell()))

    def readrecord(self):
        return tuple(self._line().strip() for _ in range(4))

    def readrecordat(self, hit):
        """the four lines of the record a hit lies in: its bases stand in front of the '+' line, so the record found
        from the hit's position is the one before"""
        self.fd.seek(hit.file_pos)
        self.seekback()
        self.readrecord()
        return '\n'.join(self.readrecord()) + '\n'

    def readhit(self, hit):
        """the bases of a hit as they stand in the file (``file_pos`` is where the trimmed read begins; a hit that
        starts inside the read has a negative ``seq_pos``)"""
        self.fd.seek(hit.file_pos + max(0, -hit.seq_pos))
        return self.fd.read(hit.length).decode('latin-1')

    def readhits(self, hits):
        return list(map(self.readhit, hits))
