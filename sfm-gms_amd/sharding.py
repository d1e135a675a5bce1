"""Pair scheduling over ranks: image pairs are independent, so the pair list
[0, P) is cut into contiguous blocks, one per GPU/process. No collective on the data path."""
import math


def all_pairs_count(n_frames):
    """N(N-1)/2 unordered pairs (a < b) of an n_frames sequence (BASELINE config 3)."""
    return n_frames * (n_frames - 1) // 2


def _row_start(a, n):
    return a * (2 * n - a - 1) // 2


def pair_from_index(k, n_frames):
    """k-th pair (a, b), a < b, in lexicographic order."""
    if not 0 <= k < all_pairs_count(n_frames):
        raise IndexError(k)
    n = n_frames
    a = (2 * n - 1 - math.isqrt((2 * n - 1) ** 2 - 8 * k)) // 2
    a = max(0, min(a, n - 2))
    while a > 0 and k < _row_start(a, n):
        a -= 1
    while k >= _row_start(a + 1, n):
        a += 1
    return a, a + 1 + (k - _row_start(a, n))


def shard_range(n_items, rank, world_size):
    """Contiguous block [lo, hi) of rank; sizes differ by at most one; blocks tile [0, n_items)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
