"""TEST INFRASTRUCTURE — numpy restatement of the vertex-range sharding plan of csrc/dist.hip (build_plan), against which
the C++ plan is checked array for array (tests/test_sharded_cpu.py). Never imported by the product.

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


def chunk_rows(row_ptr_local, n_chunks):
    """local row boundaries of n_chunks chunks balanced on sum(deg + 2) (chunks may be empty for tiny shards)"""
    rp = np.asarray(row_ptr_local, dtype=np.int64)
    n = len(rp) - 1
    c = np.cumsum(np.diff(rp) + 2)
    cuts = [0]
    for k in range(1, n_chunks):
        b = int(np.searchsorted(c, c[-1] * k / n_chunks)) + 1 if n else 0
        cuts.append(min(max(b, cuts[-1]), n))
    cuts.append(n)
    return np.array(cuts, dtype=np.int64)


class ShardPlan:
    """everything shard `rank` needs, in local indices. With n_chunks > 1 the owned rows are cut into
    chunks so the new marginals of chunk c can travel while chunk c+1 is being swept; because the halo
    is sorted by global id, the entries a peer sends for one of ITS chunks are a contiguous slice."""

    def __init__(self, row_ptr, nbr, bounds, rank, n_chunks=1):
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
        # ---- chunked views of the same lists
        self.n_chunks = max(1, int(n_chunks))
        W, Cn = self.world, self.n_chunks
        all_chunks = [bounds[p] + chunk_rows(row_ptr[bounds[p]:bounds[p + 1] + 1] - row_ptr[bounds[p]], Cn) for p in range(W)]
        self.chunk_row = (all_chunks[rank] - lo).astype(np.uint32)
        send_off = np.concatenate([[0], np.cumsum(self.send_counts)])
        self.send_counts_cp = np.zeros((Cn, W), dtype=np.int64)
        pieces = [[None] * W for _ in range(Cn)]
        for p in range(W):
            rows_p = self.send_idx[send_off[p]:send_off[p + 1]]
            cut = np.searchsorted(rows_p, self.chunk_row.astype(np.int64))
            for c in range(Cn):
                pieces[c][p] = rows_p[cut[c]:cut[c + 1]]
                self.send_counts_cp[c, p] = cut[c + 1] - cut[c]
        self.send_idx_chunked = np.concatenate([pieces[c][p] for c in range(Cn) for p in range(W)]) if self.send_idx.size else self.send_idx
        self.send_off_cp = np.concatenate([[0], np.cumsum(self.send_counts_cp.ravel())])[:-1].reshape(Cn, W)
        halo_off = np.concatenate([[0], np.cumsum(self.recv_counts)])
        self.recv_counts_cp = np.zeros((Cn, W), dtype=np.int64)
        self.recv_off_cp = np.zeros((Cn, W), dtype=np.int64)  # offsets into the halo part of the table
        for p in range(W):
            h = remote[halo_off[p]:halo_off[p + 1]]
            cut = np.searchsorted(h, all_chunks[p])
            for c in range(Cn):
                self.recv_counts_cp[c, p] = cut[c + 1] - cut[c]
                self.recv_off_cp[c, p] = halo_off[p] + cut[c]
        # The halo part of the table is kept in RECEIVE order, (chunk, peer, global id): what arrives for one chunk is one
        # contiguous slice (a single all_to_all_single), and a staged row r IS halo row r - the sweep kernel can gather
        # halo marginals straight from the receive buffer. Rename the halo entries accordingly.
        self.stage_off_c = np.concatenate([[0], np.cumsum(self.recv_counts_cp.sum(1))])
        pieces = [np.arange(self.recv_off_cp[c, p], self.recv_off_cp[c, p] + self.recv_counts_cp[c, p]) for c in range(Cn) for p in range(W)]
        by_receive = (np.concatenate(pieces) if pieces else np.zeros(0)).astype(np.int64)  # peer-major index of staged row r
        if self.n_halo:
            renamed = np.empty(self.n_halo, dtype=np.int64)
            renamed[by_receive] = np.arange(self.n_halo, dtype=np.int64)
            self.halo_global = remote[by_receive]
            halo_edges = self.nbr_local >= self.n_own
            self.nbr_local[halo_edges] = (self.n_own + renamed[self.nbr_local[halo_edges].astype(np.int64) - self.n_own]).astype(np.uint32)
            off = 0
            for c in range(Cn):
                for p in range(W):
                    self.recv_off_cp[c, p] = off
                    off += int(self.recv_counts_cp[c, p])
        self.stage_to_halo = np.arange(self.n_halo, dtype=np.int64)  # identity (kept for the unpack interface)
        self.send_off_c = np.concatenate([[0], np.cumsum(self.send_counts_cp.sum(1))])
        # send slots of every own row (CSR): where the sweep kernel drops a freshly computed marginal for its readers
        rows = self.send_idx_chunked.astype(np.int64)
        self.snd_slot = np.argsort(rows, kind="stable").astype(np.uint32)
        self.snd_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=self.n_own))]).astype(np.uint32)

    def summary(self):
        return dict(rank=self.rank, n_own=self.n_own, n_halo=self.n_halo, n_edges=self.n_edges,
                    n_send=int(self.send_counts.sum()), cut_fraction=float((self.nbr_local >= self.n_own).mean()) if self.n_edges else 0.0)
