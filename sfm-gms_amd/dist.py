"""Multi-GPU driver logic: one process per GPU, pairs sharded, no collective on the data path
(SURVEY.md section 8e -- image pairs are independent calls of the reference's matchGMS).

torch.distributed is only the rendezvous: a barrier around the timed region and a MAX over the ranks'
wall times (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
import os

from .sharding import all_pairs_count, pair_from_index, shard_range


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def rank_pair_indices(n_total_pairs, rank, world_size):
    """Global indices [lo, hi) of the pairs this rank filters."""
    return shard_range(n_total_pairs, rank, world_size)


def sequence_pair_table(n_frames, lo, hi, match_stride, stride_walk=7919, offset=0):
    """(frame_a, frame_b, m, match_off) rows for global pair indices [lo, hi) of an n_frames sequence.
    Pair k of the job is the ((k * stride_walk + offset) mod P)-th pair of the sequence, so that a shard
    touches many different frames; match_off is local to the shard."""
    total = all_pairs_count(n_frames)
    rows = []
    for k in range(lo, hi):
        a, b = pair_from_index((k * stride_walk + offset) % total, n_frames)
        rows.append((a, b, match_stride, (k - lo) * match_stride))
    return rows


def max_over_ranks(value, dist=None, device=None):
    """MAX-reduce a python float over the ranks (identity when dist is None)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(value, dist=None, device=None):
    """All ranks' integer `value`, as a list ordered by rank (for concatenating per-rank results)."""
    if dist is None:
        return [int(value)]
    import torch
    world = dist.get_world_size()
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(o.item()) for o in out]
