import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import tempfile
import numpy as np
import cases
from kvarq_amd import scan
name = sys.argv[1]
case = cases.by_name()[name]
tmp = tempfile.mkdtemp()
files = case.materialize(tmp)
data = np.frombuffer(open(files[0], 'rb').read(), dtype=np.uint8)
t = scan.Table(case.seq_bytes(), **case.config)
print('seeded:', sum(t.seeded), 'of', t.nseq, 'bytes', data.nbytes)
res = []
for force in (False, True):
    s = scan.Scanner(t); s.force_exhaustive(force); s.scan_host(data); res.append(s.finish()); s.close()
a, b = res
print(a['path'], 'records', a['stats']['records_parsed'], b['stats']['records_parsed'], 'hits', len(a['hits']), len(b['hits']))
ra, rb = a['stats']['readlengths'], b['stats']['readlengths']
print('len', len(ra), len(rb))
for i in range(max(len(ra), len(rb))):
    x = ra[i] if i < len(ra) else None; y = rb[i] if i < len(rb) else None
    if x != y: print('  rl', i, x, y)
sa, sb = set(a['hits']), set(b['hits'])
print('only seeded', sorted(sa - sb)[:10]); print('only exhaustive', sorted(sb - sa)[:10])
