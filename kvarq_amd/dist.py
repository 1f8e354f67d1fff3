"""
Read-sharded scanning over the GPUs of one node.

Records are independent, so every rank scans its own contiguous share of the
input with no data-path collective; the only exchange is one sum all-reduce of
the flat counter array (records, read-length histogram, per-sequence hit
counters, coverage and mutation counts; include/kvarq_hip.h) plus a max for the
longest-read slot, over RCCL/xGMI (``torch.distributed`` backend ``nccl``) --
or ``gloo`` on CPUs in the tests.  Hit lists are variable-length and stay on
their rank (gather them with ``gather_hits`` when a caller wants them all).
"""
import ctypes as C

from . import _lib


class NativeComm(object):
    """an RCCL communicator made by libkvarq_hip.so itself (include/kvarq_hip.h, "several GPUs"): what a C caller
    of the library uses; ``Scanner.set_comm`` makes ``finish`` sum the counters of all ranks over it and
    ``Scanner.gather_hits`` collects the hit arrays.  `exchange(id_or_None) -> id`: how rank 0's 128-byte id
    reaches the other ranks (``from_torch`` broadcasts it with torch.distributed)."""

    def __init__(self, world, rank, exchange):
        L = _lib.lib()
        buf = C.create_string_buffer(128)
        if rank == 0 and L.kvq_comm_unique_id(buf) != 0:
            raise RuntimeError(_lib.last_error()[1])
        uid = exchange(buf.raw if rank == 0 else None)
        self.h = L.kvq_comm_create(world, rank, C.create_string_buffer(bytes(uid), 128))
        if not self.h:
            raise RuntimeError(_lib.last_error()[1])
        self.world, self.rank = world, rank

    @classmethod
    def from_torch(cls, dist, device=None):
        import torch

        def exchange(uid):
            t = torch.zeros(128, dtype=torch.uint8, device=device) if uid is None else torch.tensor(list(uid), dtype=torch.uint8, device=device)
            dist.broadcast(t, src=0)
            return bytes(t.cpu().tolist())
        return cls(dist.get_world_size(), dist.get_rank(), exchange)

    def close(self):
        if self.h:
            _lib.lib().kvq_comm_destroy(self.h)
            self.h = None


class LocalComm(object):
    """the loopback communicator of libkvarq_hip.so (``kvq_comm_create_local``): the ranks are threads of this one
    process, each with a ``Scanner`` of its own, and exchange through host memory.  `key`: any number the ranks of
    one communicator share.  The join code behind it is the one RCCL communicators use -- this is how it runs with
    more than one rank on a box with a single GPU."""

    def __init__(self, world, rank, key):
        self.h = _lib.lib().kvq_comm_create_local(world, rank, key)
        if not self.h:
            raise RuntimeError(_lib.last_error()[1])
        self.world, self.rank = world, rank

    def close(self):
        if self.h:
            _lib.lib().kvq_comm_destroy(self.h)
            self.h = None


def shard(n_items, rank, world):
    """contiguous share [begin, end) of n_items for `rank`; shares differ by at most one item"""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_counters(ctr, dist, group=None):
    """in-place all-reduce of a counter tensor laid out as include/kvarq_hip.h describes:
    everything is summed, slot CTR_LONGEST (longest read + 1) is max-reduced"""
    longest = ctr[_lib.CTR_LONGEST].clone()
    dist.all_reduce(ctr, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(longest, op=dist.ReduceOp.MAX, group=group)
    ctr[_lib.CTR_LONGEST] = longest
    return ctr


def gather_hit_arrays(arrays, dist, group=None, device=None):
    """every rank's hit arrays (``Scanner.hit_arrays``) in rank order = stream order (ranks scan consecutive
    stretches, csrc/workhorse.c:1417-1431 builds one list): a count exchange, then one all-gather per array,
    padded to the largest rank -- tensors, not pickled objects, so it runs over RCCL on device memory as well as
    over gloo.  Returns the merged arrays (hitseq offsets rebased onto the merged bytes)."""
    import numpy as np
    import torch
    world = dist.get_world_size(group)
    n, nb = len(arrays['seq_nr']), len(arrays['blob'])
    counts = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n, nb], dtype=torch.int64, device=device), group=group)
    counts = [c.cpu().tolist() for c in counts]
    out = {}
    for key, width in (('seq_nr', 0), ('file_pos', 0), ('seq_pos', 0), ('length', 0), ('readlength', 0), ('blob', 1)):
        m = max(c[width] for c in counts)
        a = arrays[key]
        pad = torch.zeros(max(m, 1), dtype=torch.from_numpy(a[:0].copy()).dtype, device=device)
        pad[:len(a)] = torch.from_numpy(np.ascontiguousarray(a)).to(pad.device)
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out[key] = np.concatenate([p[:c[width]].cpu().numpy() for p, c in zip(parts, counts)])
    lens = np.maximum(out['length'], 0).astype(np.int64)
    out['offsets'] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return out


def hits_from_arrays(arrays):
    """(hits, hitseqs) in the reference's shape from merged hit arrays"""
    from .engine import Hit
    n = len(arrays['seq_nr'])
    hits = tuple(Hit(int(arrays['seq_nr'][i]), int(arrays['file_pos'][i]), int(arrays['seq_pos'][i]), int(arrays['length'][i]),
                     int(arrays['readlength'][i])) for i in range(n))
    blob, off = arrays['blob'].tobytes(), arrays['offsets']
    return hits, [blob[off[i]:off[i + 1]] for i in range(n)]
