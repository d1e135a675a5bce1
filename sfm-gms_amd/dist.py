"""Multi-GPU driver of BASELINE config 3: one process per GPU, the pair list of ONE sequence sharded over the ranks,
no collective on the data path (SURVEY.md section 8e -- image pairs are independent calls of the reference's matchGMS,
FeatureMatchUtil.cpp:66-69 once per pair).

Everything bench.py does per rank that is not the filter itself lives here, so that the world-size-2 gloo test
(tests/test_host.py) runs the very functions the benchmark runs:

  * the global pair list: all N(N-1)/2 pairs (a < b) of the sequence in lexicographic order; rank r owns the contiguous
    block shard_range(P, r, world) and walks it in chunks (one chunk = one step of the benchmark);
  * the putative matches of global pair k: a pure function of (k, match index) -- a counter hash, no generator state --
    so every rank, every world size and the host (numpy) and device (torch) forms produce the same bytes;
  * the parity sample: every 997th global pair, whichever rank filtered it;
  * the rendezvous: torch.distributed over gloo on CPU tensors -- a barrier around the timed region, a MAX over the
    ranks' wall times, a SUM of the parity counts. The path needs no RCCL.
"""
import os

import numpy as np

from .sharding import all_pairs_count, shard_range
from .types import DMATCH_DTYPE, PAIR_DTYPE

PARITY_EVERY = 997
MATCH_SEED = 0x5F3759DF
_M32 = 0xFFFFFFFF


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


# ---- the global pair list ---------------------------------------------------------------------------------------------
def pairs_from_indices(k, n_frames):
    """Vectorised sharding.pair_from_index: (a, b) arrays, a < b, of the k-th pairs in lexicographic order."""
    k = np.asarray(k, dtype=np.int64)
    n = int(n_frames)
    if k.size and (k.min() < 0 or k.max() >= all_pairs_count(n)):
        raise IndexError("pair index out of range")
    t = 2 * n - 1
    a = ((t - np.sqrt(np.maximum((t * t - 8 * k).astype(np.float64), 0.0))) / 2).astype(np.int64)
    a = np.clip(a, 0, max(n - 2, 0))
    row = lambda x: x * (2 * n - x - 1) // 2
    a = np.where(k < row(a), a - 1, a)          # the float square root may land one row off either way
    a = np.where(k >= row(a + 1), a + 1, a)
    return a, a + 1 + (k - row(a))


def pair_table(n_frames, lo, hi, m):
    """gms_pair rows (PAIR_DTYPE) for global pairs [lo, hi): m matches each, match_off local to the block."""
    k = np.arange(lo, hi, dtype=np.int64)
    a, b = pairs_from_indices(k, n_frames)
    t = np.zeros(len(k), dtype=PAIR_DTYPE)
    t["frame_a"], t["frame_b"], t["m"] = a, b, m
    t["match_off"] = (k - lo) * m
    return t


def chunk_starts(lo, hi, chunk, n_chunks):
    """Global index of the first pair of each of n_chunks successive chunks of a rank's block [lo, hi): the block is
    walked front to back and starts over when it is used up (a chunk never straddles the end)."""
    span = hi - lo
    if span <= 0 or chunk <= 0:
        raise ValueError("empty shard")
    chunk = min(chunk, span)
    per_lap = span // chunk
    return [lo + (c % per_lap) * chunk for c in range(n_chunks)], chunk


def parity_sample(first, count, every=PARITY_EVERY):
    """Global pair indices of [first, first + count) that the parity check looks at."""
    k0 = ((first + every - 1) // every) * every
    return list(range(k0, first + count, every))


# ---- putative matches of a global pair ----------------------------------------------------------------------------------
def _mix_np(x):
    x = x & _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def synth_matches_host(pair_index, n_kp, inlier_frac, seed=MATCH_SEED):
    """Matches of global pair `pair_index` (DMATCH_DTYPE, M = n_kp, queryIdx = i like BFMatcher without cross-check,
    FeatureMatchUtil.cpp:66-68): trainIdx = i with probability inlier_frac (keypoint i of every frame of
    synth.make_sequence observes the same scene point), uniformly random otherwise; distance uniform in [0, 256)."""
    i = np.arange(n_kp, dtype=np.uint64)
    key = np.uint64(pair_index) * np.uint64(n_kp) + i
    x = _mix_np((key & np.uint64(_M32)) ^ _mix_np((key >> np.uint64(32)) + np.uint64(seed)))
    r_in, r_t, r_d = _mix_np(x + np.uint64(1)), _mix_np(x + np.uint64(2)), _mix_np(x + np.uint64(3))
    thresh = np.uint64(min(int(inlier_frac * 4294967296.0), _M32))
    m = np.zeros(n_kp, dtype=DMATCH_DTYPE)
    m["queryIdx"] = i.astype(np.int32)
    m["trainIdx"] = np.where(r_in < thresh, i, r_t % np.uint64(n_kp)).astype(np.int32)
    m["imgIdx"] = 0
    m["distance"] = ((r_d >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 65536.0))
    return m


def _mix_t(x):
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32   # wraps in int64; the low 32 bits are what is kept
    x = x ^ (x >> 16)
    return x


def synth_matches_device(first, count, n_kp, inlier_frac, device, seed=MATCH_SEED, out=None):
    """The matches of global pairs [first, first + count) generated on `device` (torch), pair j of the block at rows
    [j * n_kp, (j + 1) * n_kp): an int32 [count * n_kp, 4] tensor laid out like gms_dmatch. Same bytes as
    synth_matches_host for every pair."""
    import torch
    total = count * n_kp
    if out is None:
        out = torch.empty((total, 4), dtype=torch.int32, device=device)
    thresh = min(int(inlier_frac * 4294967296.0), _M32)
    step = max(1, (1 << 24) // n_kp)  # pairs per slice: bounds the int64 temporaries
    for j0 in range(0, count, step):
        j1 = min(count, j0 + step)
        key = torch.arange((first + j0) * n_kp, (first + j1) * n_kp, dtype=torch.int64, device=device)
        i = key - (torch.arange(first + j0, first + j1, dtype=torch.int64, device=device) * n_kp).repeat_interleave(n_kp)
        x = _mix_t((key & _M32) ^ _mix_t((key >> 32) + seed))
        r_in, r_t, r_d = _mix_t(x + 1), _mix_t(x + 2), _mix_t(x + 3)
        blk = out[j0 * n_kp:j1 * n_kp]
        blk[:, 0] = i.to(torch.int32)
        blk[:, 1] = torch.where(r_in < thresh, i, r_t % n_kp).to(torch.int32)
        blk[:, 2] = 0
        blk[:, 3] = ((r_d >> 8).to(torch.float32) * (1.0 / 65536.0)).view(torch.int32)
    return out


def synth_descriptors_device(n_frames, n_kp, kind, outlier_frac, device, seed=MATCH_SEED):
    """Descriptors of every keypoint of a synth.make_sequence()-style sequence, generated on `device`: scene point i has a base
    descriptor, frame f observes it with noise, and with probability outlier_frac shows something else entirely (so that
    brute-force matching yields true correspondences i -> i mixed with arbitrary ones, like real putative matches).
    kind "orb": uint8 [n_frames * n_kp, 32] (about 32 of the 256 bits flipped per observation);
    kind "sift": float32 [n_frames * n_kp, 128] holding integers 0..255, as cv::SIFT emits them."""
    import torch
    words = 8 if kind == "orb" else 128
    thresh = min(int(outlier_frac * 4294967296.0), _M32)
    out = []
    point = torch.arange(n_kp, dtype=torch.int64, device=device)
    w = torch.arange(words, dtype=torch.int64, device=device)
    base = _mix_t(_mix_t(point[:, None] * 131 + seed) + w[None, :] * 0x9E3779B1)
    for f in range(n_frames):
        key = _mix_t(_mix_t(point[:, None] + (f + 1) * 0x10001 + seed) + w[None, :] * 0x85EBCA6B)
        other = (_mix_t(point * 7919 + f * 104729 + seed) < thresh)[:, None]
        if kind == "orb":
            noise = key & _mix_t(key + 1) & _mix_t(key + 2)              # each bit set with probability 1/8
            d = torch.where(other, _mix_t(key + 3), base ^ noise)
            out.append(d.to(torch.int32))                                # the low 32 bits, reinterpreted
        else:
            b = (base & 0xFF) * (_mix_t(base + 5) & 0xFF) >> 9           # skewed towards small values, 0..127
            n = (key & 7) - (_mix_t(key + 1) & 7)                        # -7..7
            d = torch.where(other, (_mix_t(key + 3) & 0xFF) * (_mix_t(key + 4) & 0xFF) >> 9, torch.clamp(b + n, 0, 255))
            out.append(d.to(torch.float32))
    t = torch.cat(out)
    return t.view(torch.uint8).reshape(-1, 32) if kind == "orb" else t


# ---- rendezvous ---------------------------------------------------------------------------------------------------------
def init_rendezvous(world_size):
    """torch.distributed over gloo (CPU tensors) when world_size > 1, else None. MASTER_ADDR/PORT, RANK, WORLD_SIZE come
    from the launcher's environment."""
    if world_size <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo")
    return dist


def barrier(dist=None):
    if dist is not None:
        dist.barrier()


def max_over_ranks(value, dist=None):
    """MAX-reduce a python float over the ranks (identity when dist is None)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, dist=None):
    """SUM-reduce a list of python ints over the ranks."""
    if dist is None:
        return [int(v) for v in values]
    import torch
    t = torch.tensor([int(v) for v in values], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def gather_counts(value, dist=None):
    """All ranks' integer `value`, as a list ordered by rank (for concatenating per-rank results)."""
    if dist is None:
        return [int(value)]
    import torch
    world = dist.get_world_size()
    t = torch.tensor([int(value)], dtype=torch.int64)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(o.item()) for o in out]


class RankPlan:
    """What one rank of the config-3 job does: its block of the global pair list and the chunks it walks."""

    def __init__(self, n_frames, n_kp, pairs_per_step, n_chunks, rank, world_size):
        self.n_frames, self.n_kp = int(n_frames), int(n_kp)
        self.rank, self.world = int(rank), int(world_size)
        self.total_pairs = all_pairs_count(self.n_frames)
        self.lo, self.hi = shard_range(self.total_pairs, self.rank, self.world)
        self.starts, self.chunk = chunk_starts(self.lo, self.hi, int(pairs_per_step), int(n_chunks))

    def chunk_pairs(self, c):
        """PAIR_DTYPE table of chunk c (match_off local to the chunk's match array)."""
        return pair_table(self.n_frames, self.starts[c], self.starts[c] + self.chunk, self.n_kp)

    def chunk_sample(self, c):
        """[(global pair index, index inside chunk c)] of the chunk's parity sample."""
        return [(k, k - self.starts[c]) for k in parity_sample(self.starts[c], self.chunk)]
