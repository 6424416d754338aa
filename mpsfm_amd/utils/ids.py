"""Point-id bookkeeping of the problem assembly."""

from __future__ import annotations

import numpy as np


def unique_ids(ids: np.ndarray, return_counts: bool = False):
    """``np.unique(ids, return_inverse=True[, return_counts=True])`` for non-negative integer ids.

    COLMAP point ids are small consecutive integers, so a presence table (O(n + max id)) replaces the sort
    (O(n log n): 6 ms per 150 k ids, twice per ``Optimizer.ba()`` call); sparse id ranges fall back to NumPy."""
    ids = np.asarray(ids)
    n = len(ids)
    if n == 0 or int(ids.max()) > 8 * n + 1024:
        return np.unique(ids, return_inverse=True, return_counts=return_counts)
    idx = ids.astype(np.int64)
    counts = np.bincount(idx)
    present = counts > 0
    uniq = np.flatnonzero(present)
    inverse = (np.cumsum(present) - 1)[idx]
    if return_counts:
        return uniq.astype(ids.dtype), inverse, counts[uniq]
    return uniq.astype(ids.dtype), inverse


def index_in_sorted(sorted_ids: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """``np.searchsorted(sorted_ids, ids)`` for ids that all occur in ``sorted_ids`` (unique, ascending): a lookup table
    when the id range is dense, the binary search otherwise."""
    sorted_ids, ids = np.asarray(sorted_ids), np.asarray(ids)
    n = len(sorted_ids)
    if n == 0 or len(ids) == 0 or int(sorted_ids[-1]) > 8 * n + 1024:
        return np.searchsorted(sorted_ids, ids)
    lut = np.zeros(int(sorted_ids[-1]) + 1, np.int64)
    lut[sorted_ids.astype(np.int64)] = np.arange(n)
    return lut[ids.astype(np.int64)]
