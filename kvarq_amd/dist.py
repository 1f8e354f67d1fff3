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
from . import _lib


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


def gather_hits(hits, dist, group=None):
    """all ranks' hit lists merged in canonical (stream) order: file_pos is global, so a
    sort by (file_pos, seq_nr) of the per-rank canonical lists restores the reference's order
    (ties inside one read keep their per-rank order: Python's sort is stable)"""
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, list(hits), group=group)
    merged = [h for p in parts for h in p]
    merged.sort(key=lambda h: (h[1], h[0]))
    return merged
