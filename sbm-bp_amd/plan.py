"""Vertex-range sharding plan (host logic, numpy only).

Each shard owns a contiguous row range, chosen so that sum(deg + 2) is balanced; its marginal
table is [owned rows | halo vertices], halo vertices sorted by global id (hence grouped by owner,
in owner order). Because the graph is symmetric, "vertices of mine that peer p reads" equals
"my vertices that have a neighbour in p's range", so every rank derives its send lists from its
own rows alone and they line up with the receivers' halo order without any negotiation.
"""
import numpy as np


def partition_rows(row_ptr, world):
    """bounds[r]..bounds[r+1] = rows of shard r, balancing sum(deg + 2)"""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    n = len(row_ptr) - 1
    w = np.diff(row_ptr) + 2
    c = np.cumsum(w)
    bounds = [0]
    for r in range(1, world):
        b = int(np.searchsorted(c, c[-1] * r / world)) + 1
        b = min(max(b, bounds[-1] + 1), n - (world - r))  # every shard keeps at least one row
        bounds.append(b)
    bounds.append(n)
    return np.array(bounds, dtype=np.int64)


class ShardPlan:
    """everything shard `rank` needs, in local indices"""

    def __init__(self, row_ptr, nbr, bounds, rank):
        row_ptr = np.asarray(row_ptr, dtype=np.int64)
        bounds = np.asarray(bounds, dtype=np.int64)
        self.rank, self.world = int(rank), len(bounds) - 1
        self.bounds = bounds
        self.n_global = len(row_ptr) - 1
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        self.row0, self.n_own = lo, hi - lo
        e_lo, e_hi = int(row_ptr[lo]), int(row_ptr[hi])
        self.edge0, self.n_edges = e_lo, e_hi - e_lo
        self.row_ptr = (row_ptr[lo:hi + 1] - e_lo).astype(np.uint64)
        nb = np.asarray(nbr[e_lo:e_hi], dtype=np.int64)
        own = (nb >= lo) & (nb < hi)
        remote = np.unique(nb[~own])  # ascending global ids == grouped by owner
        self.halo_global = remote
        self.n_halo = len(remote)
        owner = np.searchsorted(bounds, remote, side="right") - 1
        self.recv_counts = np.bincount(owner, minlength=self.world).astype(np.int64)
        nbr_local = np.where(own, nb - lo, 0)
        if self.n_halo:
            nbr_local[~own] = self.n_own + np.searchsorted(remote, nb[~own])
        self.nbr_local = nbr_local.astype(np.uint32)
        # send lists: my rows with a neighbour owned by p, ascending, for p = 0..world-1
        src = np.repeat(np.arange(self.n_own, dtype=np.int64), np.diff(self.row_ptr.astype(np.int64)))
        dst_owner = np.searchsorted(bounds, nb[~own], side="right") - 1
        key = np.unique(dst_owner * np.int64(max(self.n_own, 1)) + src[~own])
        self.send_counts = np.bincount(key // max(self.n_own, 1), minlength=self.world).astype(np.int64)
        self.send_idx = (key % max(self.n_own, 1)).astype(np.int64)  # local row ids, grouped by destination
        self.deg = np.diff(self.row_ptr.astype(np.int64))

    def summary(self):
        return dict(rank=self.rank, n_own=self.n_own, n_halo=self.n_halo, n_edges=self.n_edges,
                    n_send=int(self.send_counts.sum()), cut_fraction=float((self.nbr_local >= self.n_own).mean()) if self.n_edges else 0.0)
