"""``kvarq.fastq`` counterpart, as far as the engine needs it: the exception the
scan raises on malformed records (reference kvarq/fastq.py; resolved by the C
engine at import, csrc/workhorse.c:1598-1600) and the PHRED -> ASCII helper the
callers use to derive ``Amin`` (kvarq/fastq.py:41-42, 245-247)."""


class FastqFileFormatException(Exception):
    """raised when a record does not start with '@' or its 3rd line not with '+'"""


# quality characters in ASCII order, as kvarq/fastq.py:41-42 lists them
ASCII = ''.join(chr(c) for c in range(33, 127))

# vendor variants: name -> dQ (offset of Q=0 inside ASCII), kvarq/fastq.py:44-53
VARIANTS = {'Sanger': 0, 'Solexa': 31, 'Illumina 1.3+': 31, 'Illumina 1.5+': 31, 'Illumina 1.8+': 0}


def Q2A(Q, variant='Sanger'):
    """ASCII character of PHRED score ``Q`` (kvarq/fastq.py:245-247); Q=13 on
    Sanger/Illumina 1.8+ is '.', the product default ``Amin`` (kvarq/config.py:3)"""
    return ASCII[Q + VARIANTS[variant]]


# --- the probe -----------------------------------------------------------------

import collections
import gzip
import math
import os

from .log import lo

VendorProperties = collections.namedtuple('VendorProperties', ['Qrange', 'dQ'])


class Fastq(object):
    """
    ``kvarq.fastq.Fastq`` for Python 3 (reference kvarq/fastq.py:38-387): opens a ``.fastq`` /
    ``.fastq.gz`` file, checks the format of a sample of records, derives the PHRED offset
    (``dQ``, ``Azero``) from the score characters it sees, estimates read length and record
    count and finds the second file of a ``_1`` / ``_2`` pair.  The scan itself never goes
    through this class; callers use it to pick ``Amin`` (``Q2A``) and the file list
    (``filenames``) for ``engine.findseqs`` (kvarq/analyse.py:336-358, kvarq/cli.py:73-85).

    Text is handled as latin-1 ``str`` throughout (one character per byte).
    """

    ASCII = ASCII

    # declaration order decides the order of ``.variants`` (kvarq/fastq.py:47-53)
    vendor_variants = collections.OrderedDict((
        ('Sanger', VendorProperties(range(0, 50), 0)),
        ('Solexa', VendorProperties(range(-5, 41), 31)),
        ('Illumina 1.3+', VendorProperties(range(0, 41), 31)),
        ('Illumina 1.5+', VendorProperties(range(3, 42), 31)),
        ('Illumina 1.8+', VendorProperties(range(0, 62), 0)),
    ))

    def __init__(self, fname, variant=None, fd=None, paired=False, quiet=False):
        self.fname = fname
        if fname.endswith('.fastq.gz'):
            self.gz = True
        elif fname.endswith('.fastq'):
            self.gz = False
        else:
            raise FastqFileFormatException('fastq file must have extension ".fastq" or ".fastq.gz"')
        self.fd = fd if fd else (gzip.GzipFile(fname, 'rb') if self.gz else open(fname, 'rb'))

        # the second file of a pair (kvarq/fastq.py:90-98)
        self.fname2 = None
        if paired:
            cut = fname.rindex('.fastq')
            base = fname[:cut]
            if base[-2:] == '_1':
                fname2 = base[:-2] + '_2' + fname[cut:]
                if os.path.exists(fname2):
                    lo.info('including paired file "%s"' % fname2)
                    self.fname2 = fname2

        if sum(self.filesizes()) == 0:
            raise FastqFileFormatException('cannot scan empty file')

        min_pos, max_pos = self.min_max_score_check_file()
        lo.debug('min_pos=%d max_pos=%d' % (min_pos, max_pos))

        if variant and variant not in self.vendor_variants:
            raise FastqFileFormatException('unknown vendor variant "%s"' % variant)

        # variants whose score range holds everything that was seen (kvarq/fastq.py:111-118)
        fit = [(name, v.dQ) for name, v in self.vendor_variants.items()
               if (min_pos - v.dQ) in v.Qrange and (max_pos - v.dQ) in v.Qrange]
        if variant is None:
            if not fit:
                raise FastqFileFormatException('could not find any suitable fastq vendor variant')
            if len(set(dq for _, dq in fit)) > 1:
                raise FastqFileFormatException('cannot determine dQ with guessed vendor variants "%s"'
                                               % str([name for name, _ in fit]))
            self.variants = [name for name, _ in fit]
            self.dQ = fit[0][1]
        else:
            if variant not in [name for name, _ in fit]:
                lo.warning('specified vendor variant "%s" seems not to be compatible with file' % variant)
            self.variants = [variant]
            self.dQ = self.vendor_variants[variant].dQ
        self.Azero = self.ASCII[self.dQ]

        # read length of the first record, records by file size (kvarq/fastq.py:141-150)
        self.fd.seek(0)
        lines = [self._readline() for _ in range(4)]
        self.readlength = len(lines[1].strip('\r\n'))
        if self.gz:
            self.records_approx = None
        else:
            self.records_approx = os.path.getsize(self.fname) // max(1, len(''.join(lines)))
            if self.fname2 is not None:
                self.records_approx *= 2
        if not quiet:
            if self.gz:
                lo.info('gzipped fastq : readlength=? records_approx=? dQ=%d variants=%s' % (self.dQ, str(self.variants)))
            else:
                lo.info('fastq : readlength=%d records_approx=%d dQ=%d variants=%s'
                        % (self.readlength, self.records_approx, self.dQ, str(self.variants)))

    # -- files ---------------------------------------------------------------

    def filenames(self):
        return [self.fname, self.fname2] if self.fname2 is not None else [self.fname]

    def filesizes(self):
        return [os.path.getsize(f) for f in self.filenames()]

    def _readline(self):
        return self.fd.readline().decode('latin-1')

    # -- sampling ------------------------------------------------------------

    def _sample_points(self, n, points):
        """yields once per record to read: ``n`` records spread over ``points`` places of a plain
        file (all of them from the start of a gzipped one), kvarq/fastq.py:182-193"""
        self.fd.seek(0)
        for point in range(points):
            if not self.gz and point > 0:
                self.fd.seek(os.path.getsize(self.fname) * point // points)
                self.seekback()
            while n > (points - 1 - point) * n // points:
                yield point
                n -= 1

    def min_max_score_check_file(self, n=1000, points=10):
        """format check of a sample of records; smallest and largest score character seen, as
        indices into ``ASCII`` (kvarq/fastq.py:170-236)"""
        ret_min, ret_max = +999, -999
        if self.gz:
            lo.debug('gzipped fastq : scan %d points at start only' % n)
        identifier = None
        valid = set(self.ASCII)
        sampler = self._sample_points(n, points)
        for _ in sampler:
            identifier = self._readline().rstrip('\n\r')
            if not identifier:
                break
            if identifier[0] != '@':
                raise FastqFileFormatException('identifier (1st line of record) must begin with "@"')
            bases = self._readline().rstrip('\n\r')
            if not set(bases).issubset(set('AGCTN')):
                raise FastqFileFormatException('bases (2nd line of record) must contain only AGCTN')
            plus = self._readline().rstrip('\n\r')
            if not (plus == '+' or (plus[:1] == '+' and plus[1:] == identifier[1:])):
                raise FastqFileFormatException('separator (3rd line of record) must be == "+" or "+(ident)"')
            phredstr = self._readline().rstrip('\n\r')
            if not (len(bases) == len(phredstr) or (len(bases) == len(phredstr) - 1 and phredstr[-1] == '!')):
                raise FastqFileFormatException('bases must be ~ same length as phred score (2nd, 4th line)')
            if not set(phredstr).issubset(valid):
                raise FastqFileFormatException('phred score (4th line of record) must contain only "%s"' % self.ASCII)
            for x in phredstr:
                i = ord(x) - 33
                ret_min, ret_max = min(ret_min, i), max(ret_max, i)
        if identifier is not None and not identifier:
            # behind an empty line there must be nothing but empty lines
            while True:
                line = self._readline()
                if not line:
                    break
                if line.rstrip('\r\n') != '':
                    raise FastqFileFormatException('non-empty line after empty line (fpos=%d' % self.fd.tell())
        return ret_min, ret_max

    def lengths(self, Amin, n=1000, points=10):
        """quality-trimmed lengths of a sample of records (kvarq/fastq.py:266-293)"""
        if self.gz:
            lo.debug('gzipped fastq : scan %d points at start only' % n)
        out = []
        for _ in self._sample_points(n, points):
            _, _, _, scores = (self._readline().strip() for _ in range(4))
            _, length = self.cutoff(scores, Amin)
            if length >= 0:
                out.append(length)
        return out

    @staticmethod
    def cutoff(scores, Amin):
        """``pos, length`` of the longest CLOSED run of scores >= Amin (a run that reaches the
        end of the line is not counted; kvarq/fastq.py:295-308 -- the engine sees a closing
        newline and does count it, workhorse.c:1055-1068)"""
        length, best_pos, pos = -1, 0, 0
        for j, a in enumerate(scores):
            if ord(a) >= ord(Amin):
                if pos < 0:
                    pos = j
            else:
                if pos >= 0 and length < j - pos:
                    length, best_pos = j - pos, pos
                pos = -1
        return best_pos, length

    # -- PHRED arithmetic ----------------------------------------------------

    def A2Q(self, A):
        return self.ASCII.index(A) - self.dQ

    def Q2A(self, Q):
        return self.ASCII[Q + self.dQ]

    @staticmethod
    def Q2p(Q):
        return 10 ** (-.1 * Q)

    @staticmethod
    def p2Q(p):
        return int(-10 * math.log(p) / math.log(10))

    # -- records around a file position --------------------------------------

    def seekback(self):
        """moves the file position to the start of the current record (when it stands behind the
        record's '+' line) or of the previous one (when it stands in front of it), kvarq/fastq.py:331-350.

        Walks up line by line to a '+' line whose line two above starts with '@' (a score line
        may start with '+' too, but then the line two above holds bases)."""
        starts = []                                    # starts of the lines walked over, nearest first
        pos = self._line_start(self.fd.tell())
        while True:
            starts.append(pos)
            self.fd.seek(pos)
            line = self._readline()
            if line[:1] == '+' and pos > 0:
                up1 = self._line_start(pos - 1)
                up2 = self._line_start(up1 - 1) if up1 > 0 else 0
                self.fd.seek(up2)
                if self._readline()[:1] == '@' and up1 > 0:
                    self.fd.seek(up2)
                    return
            if pos == 0:
                self.fd.seek(0)
                return
            pos = self._line_start(pos - 1)

    def _line_start(self, pos):
        """start of the line that holds byte ``pos``"""
        step = 4096
        while pos > 0:
            lo_ = max(0, pos - step)
            self.fd.seek(lo_)
            chunk = self.fd.read(pos - lo_)
            k = chunk.rfind(b'\n')
            if k >= 0:
                return lo_ + k + 1
            pos = lo_
        return 0

    def readrecord(self):
        return tuple(self._readline().strip() for _ in range(4))

    def readrecordat(self, hit):
        """the four lines of the record a hit lies in (kvarq/fastq.py:374-381)"""
        self.fd.seek(hit.file_pos)
        self.seekback()
        self.readrecord()
        return '\n'.join(self.readrecord()) + '\n'

    def readhit(self, hit):
        """the bases of a hit as they stand in the file (kvarq/fastq.py:310-318)"""
        self.fd.seek(hit.file_pos - hit.seq_pos if hit.seq_pos < 0 else hit.file_pos)
        return self.fd.read(hit.length).decode('latin-1')

    def readhits(self, hits):
        return [self.readhit(hit) for hit in hits]
