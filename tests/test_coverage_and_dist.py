"""
CPU tests of the callers either side of the path: the Python-3 Coverage
(reference kvarq/analyse.py:25-185, pinned by the literal expectations of the
reference's tests/test_analyser.py:71-107) and the multi-GPU reduction helpers
(world_size-2 gloo processes, the oracle standing in for the GPU scan).
"""
import os
import sys

import numpy as np
import pytest

import cases
from kvarq_amd import _lib, dist as kdist, synth
from kvarq_amd.coverage import Coverage, Sequence
from kvarq_amd.engine import Hit
from oracle import oracle as O


def test_coverage_known_answers():
    # test_analyser.py:71-107
    cov = Coverage(Sequence('AACCGGTT'))
    cov.apply_hit(Hit(seq_nr=0, file_pos=-1, seq_pos=0, length=8, readlength=10), 'ATCCGGTTTT', on_plus_strand=True)
    assert cov.minf() == 1
    assert not cov.mixed()
    assert tuple(cov.coverage) == tuple([1] * 8)
    assert 1 in cov.mutations
    cov.deserialize(cov.serialize())
    assert tuple(cov.coverage) == tuple([1] * 8)
    assert 1 in cov.mutations
    cov.apply_hit(Hit(seq_nr=0, file_pos=-1, seq_pos=-2, length=8, readlength=10), 'AACCGGTT', on_plus_strand=True)
    cov.apply_hit(Hit(seq_nr=0, file_pos=-1, seq_pos=-1, length=8, readlength=10), 'ATCCGGTTA', on_plus_strand=True)
    assert 0.65 < cov.minf() < 0.69
    assert cov.mixed()
    fs = cov.fractions_at(1)
    assert list(fs.keys())[0] == 'T' and list(fs.values())[0] > 0.65
    assert list(fs.keys())[1] == 'A' and list(fs.values())[1] < 0.35


def test_counter_fold_equals_per_hit_fold(tmp_path):
    """Coverage filled from the fold arrays == Coverage filled hit by hit (both strands)"""
    case = cases.by_name()['spoligo_5k']
    f = case.materialize(tmp_path)[0]
    plus = [s.encode() for s in synth.SPOLIGO_SPACERS]
    seqs = synth.both_strands(plus)
    r = O.findseqs(f, seqs, fold=True, **case.config)
    n = len(plus)
    off = np.cumsum([0] + [len(s) for s in seqs])
    a = [Coverage(Sequence(p.decode())) for p in plus]
    for hit, hs in zip(r['hits'], r['hitseqs']):
        a[hit.seq_nr % n].apply_hit(hit, hs.decode(), hit.seq_nr < n)          # analyse.py:379-381
    b = [Coverage(Sequence(p.decode())) for p in plus]
    cov, mut = np.array(r['coverage']), np.array(r['mutations'])
    for k in range(n):
        for s, on_plus in ((k, True), (k + n, False)):
            b[k].add_counters(cov[off[s]:off[s + 1]], mut[6 * off[s]:6 * off[s + 1]], on_plus)
    for x, y in zip(a, b):
        assert x.coverage == y.coverage
        assert {k: sorted(v) for k, v in x.mutations.items()} == {k: sorted(v) for k, v in y.mutations.items()}
    # spoligos with mean coverage >= 2 on this fixture
    present = [k for k, c in enumerate(a) if c.mean() >= 2]
    assert present == [0, 1, 2] + list(range(7, 22)) + list(range(34, 43))


def test_shard_covers_everything_once():
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [kdist.shard(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def _rank_main(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    cfg = dict(cases.PRODUCT)
    n, L = 6000, 150
    rb = synth.record_bytes(L)
    a, b = kdist.shard(n, rank, world)
    data = synth.reads(g, a, b - a, L)
    r = O.scan_memory(data, seqs, fpos_base=a * rb, fold=True, nthreads=2, **cfg)
    nseq, bases = len(seqs), sum(map(len, seqs))
    ctr = torch.zeros(4 + 1024 + 2 * nseq + 7 * bases, dtype=torch.int64)       # include/kvarq_hip.h layout
    ctr[_lib.CTR_RECORDS] = r['stats']['records_parsed']
    ctr[_lib.CTR_LONGEST] = len(r['stats']['readlengths'])
    ctr[_lib.CTR_HITS] = len(r['hits'])
    rl = r['stats']['readlengths']
    ctr[4:4 + len(rl)] = torch.tensor(rl)
    o = 4 + 1024
    ctr[o:o + nseq] = torch.tensor(r['stats']['nseqhits'])
    ctr[o + nseq:o + 2 * nseq] = torch.tensor(r['stats']['nseqbasehits'])
    ctr[o + 2 * nseq:o + 2 * nseq + bases] = torch.tensor(r['coverage'])
    ctr[o + 2 * nseq + bases:] = torch.tensor(r['mutations'])
    kdist.reduce_counters(ctr, dist)
    # the hit lists: SoA arrays (what Scanner.hit_arrays hands out), a count exchange and one all-gather per array
    import numpy as np
    hs = r['hits']
    arrays = dict(seq_nr=np.array([h[0] for h in hs], dtype=np.int32), file_pos=np.array([h[1] for h in hs], dtype=np.int64),
                  seq_pos=np.array([h[2] for h in hs], dtype=np.int32), length=np.array([h[3] for h in hs], dtype=np.int32),
                  readlength=np.array([h[4] for h in hs], dtype=np.int32),
                  blob=np.frombuffer(b''.join(x if isinstance(x, bytes) else x.encode('latin-1') for x in r['hitseqs']), dtype=np.uint8))
    merged = kdist.gather_hit_arrays(arrays, dist)
    hits, hitseqs = kdist.hits_from_arrays(merged)
    if rank == 0:
        torch.save({'ctr': ctr, 'hits': [tuple(h) for h in hits], 'hitseqs': hitseqs}, os.path.join(tmp, 'reduced.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduction_equals_single_scan(tmp_path):
    """world_size 2 over gloo: sharded scans + reduce_counters / gather_hit_arrays == one scan of everything"""
    import torch
    import torch.multiprocessing as mp
    port = 29000 + os.getpid() % 2000
    mp.start_processes(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method='spawn')
    got = torch.load(os.path.join(str(tmp_path), 'reduced.pt'), weights_only=False)
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    whole = O.scan_memory(synth.reads(g, 0, 6000, 150), seqs, fold=True, nthreads=4, **cases.PRODUCT)
    ctr = got['ctr'].numpy()
    nseq, bases = len(seqs), sum(map(len, seqs))
    assert ctr[_lib.CTR_RECORDS] == 6000 and ctr[_lib.CTR_HITS] == len(whole['hits'])
    assert ctr[_lib.CTR_LONGEST] == len(whole['stats']['readlengths'])
    assert tuple(ctr[4:4 + len(whole['stats']['readlengths'])]) == whole['stats']['readlengths']
    o = 4 + 1024
    assert tuple(ctr[o:o + nseq]) == whole['stats']['nseqhits']
    assert tuple(ctr[o + nseq:o + 2 * nseq]) == whole['stats']['nseqbasehits']
    assert ctr[o + 2 * nseq:o + 2 * nseq + bases].tolist() == whole['coverage']
    assert ctr[o + 2 * nseq + bases:].tolist() == whole['mutations']
    assert [tuple(h) for h in got['hits']] == [tuple(h) for h in whole['hits']]
    assert [x if isinstance(x, bytes) else x.encode('latin-1') for x in got['hitseqs']] == [x if isinstance(x, bytes) else x.encode('latin-1') for x in whole['hitseqs']]


# ---- the native join (kvarq_amd/csrc/kvq_dist.hip): where the ranks' arrays go, on the CPU -------------------

def _rank_buffer(L, hits, hitseqs):
    """one rank's result buffer as kvq_scan_finish lays it out (kvq_result_layout_words)"""
    import ctypes as C
    n, blob = len(hits), b''.join(hitseqs)
    w = (C.c_uint64 * 8)()
    L.kvq_result_layout_words(n, len(blob), w)
    buf = np.zeros(int(w[7]) + 16, dtype=np.uint8)
    def put(at, arr):
        raw = np.ascontiguousarray(arr).view(np.uint8)
        buf[at:at + raw.size] = raw
    put(int(w[0]), np.array([h[1] for h in hits], dtype=np.int64))
    put(int(w[1]), np.cumsum([0] + [len(x) for x in hitseqs]).astype(np.int64))
    put(int(w[2]), np.array([h[0] for h in hits], dtype=np.int32)); put(int(w[3]), np.array([h[2] for h in hits], dtype=np.int32))
    put(int(w[4]), np.array([h[3] for h in hits], dtype=np.int32)); put(int(w[5]), np.array([h[4] for h in hits], dtype=np.int32))
    put(int(w[6]), np.frombuffer(blob, dtype=np.uint8))
    return buf, len(blob)


@pytest.mark.parametrize('world', [2, 3, 8])
def test_gather_plan_puts_every_ranks_arrays_in_stream_order(world):
    """kvq_gather_plan / kvq_gather_host (the arithmetic of kvq_scan_gather_hits, without a GPU): the oracle's single
    scan, cut into the ranks' shares by dist.shard -- uneven ones, and ranks without a hit -- comes back whole"""
    import ctypes as C
    L = _lib.lib()
    g = synth.genome()
    seqs = synth.both_strands(synth.table(g))
    nreads, rb = 3000, synth.record_bytes(150)
    whole = O.scan_memory(synth.reads(g, 0, nreads, 150), seqs, nthreads=4, **cases.PRODUCT)
    hits = [tuple(h) for h in whole['hits']]
    hs = [x if isinstance(x, bytes) else x.encode('latin-1') for x in whole['hitseqs']]
    # shares of very different sizes: rank 1 gets (nearly) nothing when there are more than two ranks
    cuts = [0] + [kdist.shard(nreads, r, world)[1] for r in range(world)]
    if world > 2:
        cuts[2] = cuts[1] + 1
    bufs, counts = [], []
    for r in range(world):
        lo, hi = cuts[r] * rb, cuts[r + 1] * rb
        idx = [i for i, h in enumerate(hits) if lo <= h[1] < hi]
        b, nb = _rank_buffer(L, [hits[i] for i in idx], [hs[i] for i in idx])
        bufs.append(b); counts += [len(idx), nb]
    assert sum(counts[0::2]) == len(hits) and (world == 2 or min(counts[0::2]) == 0 or counts[2] <= 1)
    cnt = (C.c_uint64 * (2 * world))(*counts)
    parts = (C.c_uint64 * (21 * world))(); base = (C.c_uint64 * world)(); tot = (C.c_uint64 * 4)()
    assert L.kvq_gather_plan(world, cnt, parts, base, tot) == 0
    assert tot[0] == len(hits) and tot[1] == sum(len(x) for x in hs)
    assert list(base) == list(np.cumsum([0] + counts[1::2])[:-1])
    out = np.zeros(int(tot[2]) + 16, dtype=np.uint8)
    ptrs = (C.c_void_p * world)(*[b.ctypes.data for b in bufs])
    assert L.kvq_gather_host(world, cnt, ptrs, out.ctypes.data) == 0
    want, _ = _rank_buffer(L, hits, hs)
    assert out[:int(tot[2])].tobytes() == want[:int(tot[2])].tobytes()
