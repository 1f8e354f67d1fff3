"""
The scan-level caller of the engine for Python 3: what ``kvarq.analyse.Analyser`` does between a
``Fastq`` and the JSON file (reference kvarq/analyse.py:189-435), without the genome / testsuite
model (kvarq/genes.py, testsuites/ -- out of scope, they need the reference genome).  Templates are
given directly as an ordered ``name -> Sequence`` mapping; everything downstream keeps the
reference's shapes:

* ``scan`` hands ``plus + reverse complements`` to ``engine.findseqs`` (analyse.py:350-358) and folds the
  hits into one ``Coverage`` per template (analyse.py:363-381),
* ``encode`` / ``decode`` produce / read the ``.json`` object (analyse.py:397-435, 438-530),
* ``json_dump`` writes it the way ``kvarq scan`` does (kvarq/util.py:272-294): two levels indented,
  everything deeper on one line.
"""
import collections
import json
import os
import time

from . import VERSION, engine
from .coverage import Coverage, Sequence
from .fastq import Fastq
from .log import lo


class DecodingException(Exception):
    """the data is not an encoded scan (analyse.py:187-188)"""


def _as_sequence(thing):
    if isinstance(thing, Sequence):
        return thing
    if isinstance(thing, bytes):
        thing = thing.decode('latin-1')
    if isinstance(thing, str):
        return Sequence(thing)
    bases, left, right = (tuple(thing) + (0, 0))[:3]
    return Sequence(bases.decode('latin-1') if isinstance(bases, bytes) else bases, left=left, right=right)


class Analyser(object):

    def __init__(self, spacing=25):
        self.spacing = spacing                  # flank length used when templates were cut (kvarq/config.py:9)
        self.fastq = None
        self.coverages = None                   # OrderedDict name -> Coverage
        self.hits = self.hitseqs = self.stats = self.config = None
        self.results = {}                       # testsuite interpretation is out of scope: stays empty
        self.scantime = -1

    # -- addressing (analyse.py:283-326) -----------------------------------------

    def coverage_at(self, seq_nr):
        """the coverage sequence number ``seq_nr`` of the scan maps to (plus strands first, then the
        reverse complements in the same order)"""
        return list(self.coverages.values())[seq_nr % len(self.coverages)]

    def __len__(self):
        return len(self.coverages)

    def __getitem__(self, thing):
        return self.coverage_at(thing) if isinstance(thing, int) else self.coverages[str(thing)]

    # -- scanning ------------------------------------------------------------------

    def scan(self, fastq, templates, do_reverse=True):
        """``fastq``: a :class:`kvarq_amd.fastq.Fastq`; ``templates``: ordered mapping name -> plus-strand
        template (``Sequence``, text, or ``(text, left, right)``).  May raise ``FastqFileFormatException``."""
        self.fastq = fastq
        self.fastq_filenames = fastq.filenames()
        self.fastq_sizes = fastq.filesizes()
        self.fastq_readlength = fastq.readlength
        self.fastq_records_approx = fastq.records_approx
        self.coverages = collections.OrderedDict((str(name), Coverage(_as_sequence(t))) for name, t in templates.items())
        self.config = engine.get_config()
        seqs = [c.plus_seq.bases for c in self.coverages.values()]
        if do_reverse:
            seqs += [c.minus_seq.bases for c in self.coverages.values()]
        t0 = time.time()
        ret = engine.findseqs(self.fastq_filenames, seqs)
        lo.debug('found %d hits' % len(ret['hits']))
        self.stats, self.hits, self.hitseqs = ret['stats'], ret['hits'], ret['hitseqs']
        self.scantime = time.time() - t0
        self.update_coverages()

    def update_coverages(self):
        """applies ``.hits`` to fresh coverages (analyse.py:363-381)"""
        assert self.hits is not None and self.hitseqs is not None, 'cannot update coverages without .hits / .hitseqs'
        for c in self.coverages.values():
            c.clear()
        n = len(self.coverages)
        for hit, hitseq in zip(self.hits, self.hitseqs):
            if isinstance(hitseq, bytes):
                hitseq = hitseq.decode('latin-1')
            self.coverage_at(hit.seq_nr).apply_hit(hit, hitseq, hit.seq_nr < n)

    # -- the .json object ---------------------------------------------------------

    def encode(self, hits=False):
        more = {}
        if hits:
            more['hits'] = [list(h) for h in self.hits]
            more['hitseqs'] = [h.decode('latin-1') if isinstance(h, bytes) else h for h in self.hitseqs]
        config = dict((k, v.decode('latin-1') if isinstance(v, bytes) else v) for k, v in self.config.items())
        return dict(
            analyses=self.results,
            info={
                'format': 'kvarq',
                'fastq': self.fastq_filenames,
                'size': self.fastq_sizes,
                'readlength': self.fastq_readlength,
                'records_approx': self.fastq_records_approx,
                'scantime': self.scantime,
                'when': time.asctime(time.localtime()),
                'version': VERSION,
                'config': config,
                'spacing': self.spacing,
                'testsuites': {},
            },
            stats=self.stats,
            coverages=[(name, c.serialize()) for name, c in self.coverages.items()],
            **more)

    def decode(self, templates, data):
        """restores what ``scan`` left behind from an encoded object; ``templates`` as for ``scan``
        (coverages of templates that are not listed are dropped, analyse.py:503-530)"""
        if not isinstance(data, dict) or data.get('info', {}).get('format') != 'kvarq':
            raise DecodingException('not a kvarq .json object')
        info = data['info']
        self.config = info['config']
        self.fastq_filenames, self.fastq_sizes = info['fastq'], info['size']
        self.fastq_readlength = info.get('readlength', -1)
        self.fastq_records_approx = info.get('records_approx', -1)
        self.scantime = info.get('scantime', -1)
        self.spacing = info.get('spacing', self.spacing)
        self.stats = data['stats']
        self.hits = [engine.Hit(*h) for h in data['hits']] if 'hits' in data else None
        self.hitseqs = data.get('hitseqs')
        if os.path.isfile(self.fastq_filenames[0]):
            lo.info('found .fastq file : ' + self.fastq_filenames[0])
            self.fastq = Fastq(self.fastq_filenames[0], paired=len(self.fastq_filenames) > 1, quiet=True)
        else:
            lo.info('cannot load .fastq file : ' + self.fastq_filenames[0])
            self.fastq = None
        stored = collections.OrderedDict((name, text) for name, text in data['coverages'])
        self.coverages = collections.OrderedDict()
        for name, t in templates.items():
            if str(name) not in stored:
                raise DecodingException('no coverage for template "%s"' % name)
            c = Coverage(_as_sequence(t))
            c.deserialize(stored[str(name)])
            if len(c.coverage) != len(c.plus_seq):
                raise DecodingException('coverage of "%s" does not fit its template' % name)
            self.coverages[str(name)] = c


def json_dump(data, fd, indent=2, max_indent_level=2):
    """writes ``data`` as JSON with the first ``max_indent_level`` levels indented and everything
    below on one line (the layout of kvarq/util.py:272-294, so that a coverage string or a hit stays
    on one line).  Byte for byte what the reference writes: it runs Python 2.7's ``json.JSONEncoder(indent=2)``,
    whose item separator stays ", " when it indents, so an indented line that is followed by another ends in
    a comma AND a blank; below the indented levels the separators are ", " and ": " as well."""
    def emit(obj, level):
        if level >= max_indent_level or not isinstance(obj, (dict, list, tuple)) or not obj:
            fd.write(json.dumps(obj, separators=(', ', ': ')))
            return
        pad, pad_in = ' ' * (indent * level), ' ' * (indent * (level + 1))
        if isinstance(obj, dict):
            fd.write('{\n')
            for i, (k, v) in enumerate(obj.items()):
                fd.write(pad_in + json.dumps(str(k)) + ': ')
                emit(v, level + 1)
                fd.write(', \n' if i + 1 < len(obj) else '\n')
            fd.write(pad + '}')
        else:
            fd.write('[\n')
            for i, v in enumerate(obj):
                fd.write(pad_in)
                emit(v, level + 1)
                fd.write(', \n' if i + 1 < len(obj) else '\n')
            fd.write(pad + ']')
    emit(data, 0)
