"""Multi-GPU BP: one process per GPU, vertex-range shards, torch.distributed (backend "nccl" = RCCL
over xGMI) for the two exchanges a sweep needs:

  * all-to-all of the boundary vertices' marginals, received straight into the tail of each
    shard's marginal table (the marginal-gather sweep reads remote neighbours there);
  * all-reduce of Q sums (global field) and one max (convergence hint) — and of one max for the
    exact criterion when convergence is being decided.

Everything between two collectives is a C-ABI shard step (include/sbmbp.h, sbmbp_shard_*).
All collectives are issued on the stream the engine runs on, so a batch of sweeps is queued
without host synchronisation; the device-side stop flag makes sweeps queued after the trigger
no-ops. The orchestration is written over lists of local shards so that the same code drives
one shard per rank (TorchDistComm) or several shards in one process (LocalComm: tests on one
GPU, or on CPU with a stand-in backend).
"""
import ctypes as C

import os

import numpy as np

from sbm_bp_amd.plan import ShardPlan, block_cyclic_layout, busiest_link_rows, partition_rows, permute_csr

RED_GATHER_OFFSET = 32  # SBMBP_RED_GATHER_OFFSET (include/sbmbp.h): gathered rows start behind the shard's own red[0..Q]
LEARN_FIELD_MIX, LEARN_SNAP = 0.3, 1.0  # defaults of sbmbp_set_learning_schedule (EM loop rules)
EXACT_NONEDGE_MAX = 32768  # up to this many vertices the non-edge term is the exact all-pairs sum (engine.hip: nonedge_terms)


class ShardDesc(C.Structure):
    _fields_ = [("n_global", C.c_uint32), ("n_own", C.c_uint32), ("n_halo", C.c_uint32), ("row0", C.c_uint32),
                ("n_edges", C.c_uint64), ("edge0", C.c_uint64), ("row_ptr", C.POINTER(C.c_uint64)),
                ("nbr_local", C.POINTER(C.c_uint32)), ("psi_buf0", C.c_void_p), ("psi_buf1", C.c_void_p),
                ("red_buf", C.c_void_p), ("n_chunks", C.c_uint32), ("chunk_row", C.POINTER(C.c_uint32))]


class ConvState(C.Structure):
    _fields_ = [("maxdiff", C.c_double), ("conv_iter", C.c_int), ("sweep_idx", C.c_int), ("stop", C.c_int),
                ("last_exact", C.c_int)]


class HipShardBackend:
    """one shard on one GPU: thin wrapper of the sbmbp_shard_* steps; buffers are torch tensors"""

    def __init__(self, plan, Q, dc, device, compress=True):
        import torch
        from sbm_bp_amd.capi import check, load_library
        self._check, self._lib = check, load_library()
        self.torch, self.plan, self.Q, self.dc = torch, plan, Q, dc
        # halo payload: Q-1 components per marginal (they sum to 1; the receiver restores the last one)
        self.ncomp = Q - 1 if compress else Q
        self.device = torch.device("cuda", device)
        n_tab = plan.n_own + plan.n_halo
        self.psi = torch.zeros((2, n_tab, Q), dtype=torch.float64, device=self.device)
        self.red = torch.zeros(8192, dtype=torch.float64, device=self.device)
        self.send_idx = torch.as_tensor(plan.send_idx_chunked.astype(np.int32), device=self.device)
        self.sendbuf = torch.zeros((max(1, len(plan.send_idx_chunked)), self.ncomp), dtype=torch.float64, device=self.device)
        # one receive buffer per marginal table (the exchange for table t+1 runs while the sweep reads the buffer of table
        # t), rows in halo order = (chunk, peer, id): a chunk's exchange fills one contiguous slice
        self.recvbuf = torch.zeros((2, max(1, plan.n_halo), self.ncomp), dtype=torch.float64, device=self.device)
        self.stage_to_halo = torch.as_tensor(plan.stage_to_halo.astype(np.int32), device=self.device)
        self.send_views = [self.sendbuf[int(plan.send_off_c[c]):int(plan.send_off_c[c + 1])] for c in range(plan.n_chunks)]
        self._recv_views = [[self.recvbuf[t][int(plan.stage_off_c[c]):int(plan.stage_off_c[c + 1])] for c in range(plan.n_chunks)]
                            for t in (0, 1)]
        self._row_ptr = np.ascontiguousarray(plan.row_ptr, dtype=np.uint64)
        self._nbr = np.ascontiguousarray(plan.nbr_local, dtype=np.uint32)
        self._chunk_row = np.ascontiguousarray(plan.chunk_row, dtype=np.uint32)
        d = ShardDesc(plan.n_global, plan.n_own, plan.n_halo, plan.row0, plan.n_edges, plan.edge0,
                      self._row_ptr.ctypes.data_as(C.POINTER(C.c_uint64)), self._nbr.ctypes.data_as(C.POINTER(C.c_uint32)),
                      self.psi[0].data_ptr(), self.psi[1].data_ptr(), self.red.data_ptr(), plan.n_chunks,
                      self._chunk_row.ctypes.data_as(C.POINTER(C.c_uint32)))
        h = C.c_void_p()
        self._check(self._lib.sbmbp_shard_create(C.byref(h), C.byref(d), Q, dc, device))
        self._h = h
        # run on torch's current stream so kernels and collectives are ordered without host syncs
        self._check(self._lib.sbmbp_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        # fused exchange buffers: the sweep kernel fills the send buffer and reads the halo from the receive buffers, so no
        # pack / unpack kernel runs between sweeps (SBMBP_SHARD_FUSED=0: the separate kernels, for A/B runs)
        self.fused = os.environ.get("SBMBP_SHARD_FUSED", "1") != "0"
        if self.fused:
            self._snd_ptr = np.ascontiguousarray(plan.snd_ptr, dtype=np.uint32)
            self._snd_slot = np.ascontiguousarray(plan.snd_slot, dtype=np.uint32)
            dp = C.POINTER(C.c_double)
            self._check(self._lib.sbmbp_shard_set_io(self._h, self._snd_ptr.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     self._snd_slot.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     C.cast(self.sendbuf.data_ptr(), dp), C.cast(self.recvbuf[0].data_ptr(), dp),
                                                     C.cast(self.recvbuf[1].data_ptr(), dp), self.ncomp))

    def __del__(self):
        try:
            self.torch.cuda.synchronize(self.device)
            self._lib.sbmbp_destroy(self._h)
        except Exception:
            pass

    def init_messages_device(self, seed, true_conf_local):
        tc = np.ascontiguousarray(true_conf_local, dtype=np.uint32)
        self._check(self._lib.sbmbp_init_messages_device(self._h, seed, tc.ctypes.data_as(C.POINTER(C.c_uint32))))

    def set_params(self, cab, na, beta):
        cab = np.ascontiguousarray(cab, dtype=np.float64)
        na = np.ascontiguousarray(na, dtype=np.uint32)
        self._check(self._lib.sbmbp_set_params(self._h, cab.ctypes.data_as(C.POINTER(C.c_double)),
                                               na.ctypes.data_as(C.POINTER(C.c_uint32)), beta))

    def begin(self, armed):
        self._check(self._lib.sbmbp_shard_begin(self._h, armed))

    def read_buffer(self, j):
        return self._lib.sbmbp_shard_read_buffer(self._h, j)

    def pack(self, j, c):
        """gather the chunk-c send rows of the table sweep j reads into their slice of sendbuf"""
        off = int(self.plan.send_off_cp[c, 0])
        n = int(self.plan.send_counts_cp[c].sum())
        if n:
            self._check(self._lib.sbmbp_shard_pack(self._h, j, C.cast(self.send_idx.data_ptr() + 4 * off, C.POINTER(C.c_uint32)), n,
                                                   C.cast(self.sendbuf.data_ptr() + 8 * self.ncomp * off, C.POINTER(C.c_double)),
                                                   self.ncomp))

    def recv_view(self, j, c):
        """where chunk c of the halo of the table sweep j reads is received"""
        return self._recv_views[self.read_buffer(j)][c]

    def unpack(self, j):
        """expand the received halo rows into the halo rows of the table sweep j reads"""
        if self.plan.n_halo:
            self._check(self._lib.sbmbp_shard_unpack(self._h, j, C.cast(self.recvbuf[self.read_buffer(j)].data_ptr(), C.POINTER(C.c_double)),
                                                     C.cast(self.stage_to_halo.data_ptr(), C.POINTER(C.c_uint32)),
                                                     self.plan.n_halo, self.ncomp))

    def sweep_chunk(self, j, c):
        self._check(self._lib.sbmbp_shard_sweep_chunk(self._h, j, c))

    def sweep_fold(self):
        self._check(self._lib.sbmbp_shard_sweep_fold(self._h))

    def field_partial(self, j):
        self._check(self._lib.sbmbp_shard_field_partial(self._h, j))

    def sweep_partial(self, j):
        self._check(self._lib.sbmbp_shard_sweep_partial(self._h, j))

    def finalize(self, mode, n_rows):
        self._check(self._lib.sbmbp_shard_finalize(self._h, mode, n_rows))

    def msgdiff_partial(self):
        self._check(self._lib.sbmbp_shard_msgdiff_partial(self._h))

    def rowsums_partial(self):
        self._check(self._lib.sbmbp_shard_rowsums_partial(self._h))

    def fe_partial(self, want_entropy):
        self._check(self._lib.sbmbp_shard_fe_partial(self._h, int(want_entropy)))

    def fe_finish(self):
        out = np.zeros(4)
        self._check(self._lib.sbmbp_shard_fe_finish(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def nonedge_partial(self, want_entropy):
        n, k = C.c_uint32(0), C.c_int(0)
        self._check(self._lib.sbmbp_shard_nonedge_partial(self._h, int(want_entropy), C.byref(n), C.byref(k)))
        return n.value, k.value

    def own_rows_in_global_table(self, n_global):
        """a zero (n_global, Q) table with this shard's current marginals at their global rows (to be SUM all-reduced)"""
        t = self.torch.zeros((n_global, self.Q), dtype=self.torch.float64, device=self.device)
        p = self.plan
        t[p.row0:p.row0 + p.n_own] = self.psi[self.read_buffer(0)][:p.n_own]
        return t

    def nonedge_exact_partial(self, psi_all, want_entropy):
        self._check(self._lib.sbmbp_shard_nonedge_exact_partial(self._h, C.cast(psi_all.data_ptr(), C.POINTER(C.c_double)), int(want_entropy)))

    def nonedge_finish(self, want_entropy, order):
        out = np.zeros(2)
        self._check(self._lib.sbmbp_shard_nonedge_finish(self._h, int(want_entropy), order, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def em_partial(self):
        n = C.c_uint32(0)
        self._check(self._lib.sbmbp_shard_em_partial(self._h, C.byref(n)))
        return n.value

    def em_finish(self):
        Q = self.Q
        na, nna, cab = np.zeros(Q), np.zeros(Q), np.zeros((Q, Q))
        dp = C.POINTER(C.c_double)
        self._check(self._lib.sbmbp_shard_em_finish(self._h, na.ctypes.data_as(dp), nna.ctypes.data_as(dp), cab.ctypes.data_as(dp)))
        return na, nna, cab

    def poll(self):
        st = ConvState()
        self._check(self._lib.sbmbp_shard_poll(self._h, C.byref(st)))
        return st

    def commit(self, n):
        self._check(self._lib.sbmbp_shard_commit(self._h, n))

    def set_schedule(self, field_mix, check_every=1):
        self._check(self._lib.sbmbp_set_schedule(self._h, field_mix, check_every))

    def get_state(self):
        from sbm_bp_amd.capi import c_dp
        psi = np.zeros((self.plan.n_own, self.Q))
        msg = np.zeros((self.plan.n_edges, self.Q))
        self._check(self._lib.sbmbp_get_state(self._h, psi.ctypes.data_as(c_dp), msg.ctypes.data_as(c_dp)))
        return psi, msg

    def set_state(self, psi, msg):
        from sbm_bp_amd.capi import c_dp
        psi = np.ascontiguousarray(psi, dtype=np.float64)
        msg = np.ascontiguousarray(msg, dtype=np.float64)
        self._check(self._lib.sbmbp_set_state(self._h, psi.ctypes.data_as(c_dp), msg.ctypes.data_as(c_dp)))

    def stats(self):
        from sbm_bp_amd.capi import Stats
        s = Stats()
        self._check(self._lib.sbmbp_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self._lib.sbmbp_reset_stats(self._h))

    def set_timing(self, on):
        self._check(self._lib.sbmbp_set_timing(self._h, int(on)))

    def sync(self):
        self.torch.cuda.synchronize(self.device)


class TorchDistComm:
    """one shard per rank; collectives over the default torch.distributed group"""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def local_ranks(self):
        return [self.rank]

    def exchange(self, recvs, sends, recv_counts, send_counts):
        """one all-to-all-v per call: recvs[0]/sends[0] are contiguous row blocks split by peer; returns the work"""
        if recvs[0].shape[0] == 0 and sends[0].shape[0] == 0 and self.world == 1:
            return []
        return [self.dist.all_to_all_single(recvs[0], sends[0], [int(x) for x in recv_counts[0]], [int(x) for x in send_counts[0]],
                                            async_op=True)]

    def all_reduce(self, tensors, op):
        self.dist.all_reduce(tensors[0], op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)

    def all_gather(self, outs, ins):
        self.dist.all_gather_into_tensor(outs[0], ins[0])

    def host_max(self, values):
        """element-wise max of a short list of host integers over all ranks (setup-time decisions)"""
        import torch
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [int(x) for x in t.cpu().tolist()]


class HostStagedComm(TorchDistComm):
    """rehearsal / debugging only: the same calls over a backend without device collectives (gloo), every
    buffer staged through host memory and every call blocking. RCCL refuses two ranks on one device; with
    this comm several ranks may share one GPU, so the multi-process driver can be run on a 1-GPU box.
    Never selected automatically."""

    def exchange(self, recvs, sends, recv_counts, send_counts):
        import torch
        s = sends[0].cpu()  # blocks on the stream that packed the rows
        r = torch.empty(tuple(recvs[0].shape), dtype=recvs[0].dtype)
        self.dist.all_to_all_single(r, s, [int(x) for x in recv_counts[0]], [int(x) for x in send_counts[0]])
        recvs[0].copy_(r)
        return []

    def all_reduce(self, tensors, op):
        t = tensors[0].cpu()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        tensors[0].copy_(t)

    def all_gather(self, outs, ins):
        import torch
        t = ins[0].cpu()
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        outs[0].copy_(torch.cat(parts))


class LocalComm:
    """all shards live in this process (lock-step); collectives are tensor copies"""

    def __init__(self, world):
        self.rank, self.world = 0, world

    def host_max(self, values):
        return [int(v) for v in values]

    def local_ranks(self):
        return list(range(self.world))

    def exchange(self, recvs, sends, recv_counts, send_counts):
        send_off = [np.concatenate([[0], np.cumsum(sc)]) for sc in send_counts]
        for r in range(self.world):
            off = 0
            for p in range(self.world):
                n = int(recv_counts[r][p])
                if n:
                    assert int(send_counts[p][r]) == n
                    s0 = int(send_off[p][r])
                    recvs[r][off:off + n].copy_(sends[p][s0:s0 + n])
                off += n
        return []

    def all_gather(self, outs, ins):
        import torch
        cat = torch.cat([t.to(outs[0].device) for t in ins])
        for o in outs:
            o.copy_(cat.to(o.device))

    def all_reduce(self, tensors, op):
        import torch
        stack = torch.stack([t.to(tensors[0].device) for t in tensors])
        red = stack.sum(0) if op == "sum" else stack.max(0).values
        for t in tensors:
            t.copy_(red.to(t.device))


class ShardedBP:
    """belief_propagation over vertex-range shards (infer path: converge / sweep / overlap)."""

    def __init__(self, plans, Q, dc, comm, backend_factory=None):
        self.comm, self.Q, self.dc = comm, Q, dc
        self.plans = plans
        if backend_factory is None:
            import torch
            dev = torch.cuda.current_device()
            compress = os.environ.get("SBMBP_HALO_COMPRESS", "1") != "0"  # Q-1 components on the wire (default) or all Q
            backend_factory = lambda plan: HipShardBackend(plan, Q, dc, dev, compress=compress)  # noqa: E731
        self.shards = [backend_factory(p) for p in plans]
        self.fused = all(getattr(sh, "fused", False) for sh in self.shards)
        self._timing, self._phase_log = False, []
        self.N_global = plans[0].n_global
        self.E2_local = sum(p.n_edges for p in plans)
        self.E2_global = None
        self.total_sweeps = 0

    # -- construction -------------------------------------------------------------------------
    @classmethod
    def from_csr(cls, row_ptr, nbr, Q, dc, comm, backend_factory=None, n_chunks=None, interleave=1):
        """interleave = k > 1 deals k*world row blocks round robin to the shards (plan.block_cyclic_layout) instead of
        giving each one range; "auto" picks the k in (1, 2, 4) with the lightest busiest link. The vertices are renamed
        so that every shard still owns one contiguous range: `self.order[p]` is the caller's id of vertex p, and
        true_conf / global_marginals() translate at the boundary."""
        world = comm.world
        if interleave == "auto":
            interleave = cls._choose_interleave(row_ptr, nbr, comm) if world > 1 else 1
        interleave = max(1, int(interleave))
        order = None
        if interleave > 1 and world > 1:
            order, bounds, bb = block_cyclic_layout(row_ptr, world, interleave)
            row_ptr, nbr, _ = permute_csr(row_ptr, nbr, order, bb)
        else:
            interleave = 1
            bounds = partition_rows(row_ptr, world)
        if n_chunks is None:  # SBMBP_SHARD_CHUNKS: tuning knob for the compute/exchange overlap (default 4)
            n_chunks = 1 if world == 1 else max(1, int(os.environ.get("SBMBP_SHARD_CHUNKS", "4")))
        plans = [ShardPlan(row_ptr, nbr, bounds, r, n_chunks) for r in comm.local_ranks()]
        self = cls(plans, Q, dc, comm, backend_factory)
        self.E2_global = int(len(nbr))
        self.bounds = bounds
        self.order = order          # None: the caller's numbering
        self.interleave = interleave
        return self

    @staticmethod
    def _choose_interleave(row_ptr, nbr, comm):
        """k in (1, 2, 4) minimising the rows on the busiest link of any rank"""
        cands = [k for k in (1, 2, 4) if k * comm.world <= max(1, len(row_ptr) - 1)]
        load = [max(busiest_link_rows(row_ptr, nbr, comm.world, k, r) for r in comm.local_ranks()) for k in cands]
        load = comm.host_max(load)
        best = min(range(len(cands)), key=lambda i: (load[i], cands[i]))
        # a different layout costs a larger halo: keep plain ranges unless the busiest link gets 40 % lighter
        return cands[best] if load[best] < 0.6 * load[0] else 1

    def to_caller_order(self, rows_new):
        """rows in the engine's numbering (concatenated over all shards) -> the caller's numbering"""
        if self.order is None:
            return rows_new
        out = np.empty_like(rows_new)
        out[self.order] = rows_new
        return out

    @classmethod
    def synthetic(cls, N, Q, c, eps, graph_seed, dc=0, seed=1234, comm=None):
        """every rank generates the same planted-partition graph and keeps its own row range"""
        import sbm_bp_amd as S
        from sbm_bp_amd import synth
        comm = comm or TorchDistComm()
        pairs, cin, cout = synth.planted_partition(N, Q, c, eps, graph_seed)
        g = S.Graph.from_edges(pairs, N)
        del pairs
        row_ptr, nbr, _ = g.csr()
        # plain ranges unless SBMBP_SHARD_INTERLEAVE says otherwise ("auto" or k): on the planted benchmark graphs dealing
        # blocks lightens the busiest link by at most 27 % (k = 4 at 8 shards) and pays with 1.7x the halo
        il = os.environ.get("SBMBP_SHARD_INTERLEAVE", "1")
        self = cls.from_csr(row_ptr, nbr, Q, dc, comm, interleave=il if il == "auto" else int(il))
        tc = synth.true_conf(N, Q)
        self.init_messages_device(seed, tc)
        self.expand_bp_params(synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(N, Q), dtype=np.uint32), 1.0)
        return self

    def init_messages_device(self, seed, true_conf_global):
        tc = np.asarray(true_conf_global)
        if getattr(self, "order", None) is not None:
            tc = tc[self.order]
        for sh, p in zip(self.shards, self.plans):
            sh.init_messages_device(seed, tc[p.row0:p.row0 + p.n_own])

    def expand_bp_params(self, cab, na, beta=1.0):
        self.cab = np.array(cab, dtype=np.float64)
        self.na = np.array(na, dtype=np.uint32)
        self.beta = float(beta)
        for sh in self.shards:
            sh.set_params(self.cab, self.na, self.beta)

    # -- one sweep = exchange, local sweep, reduce, finalize ------------------------------------
    def _exchange_chunk(self, j, c, packed=False):
        """ship the chunk-c boundary marginals of the table that sweep j reads: ONE all-to-all-v of contiguous slices (send
        buffer and receive buffer are ordered by (chunk, peer)). packed: the sweep kernel already filled the send buffer."""
        if not packed:
            for sh in self.shards:
                sh.pack(j, c)
        return self.comm.exchange([sh.recv_view(j, c) for sh in self.shards], [sh.send_views[c] for sh in self.shards],
                                  [p.recv_counts_cp[c] for p in self.plans], [p.send_counts_cp[c] for p in self.plans])

    def _reduce(self, n_sum, n_max):
        if n_sum:
            self.comm.all_reduce([sh.red[:n_sum] for sh in self.shards], "sum")
        if n_max:
            self.comm.all_reduce([sh.red[n_sum:n_sum + n_max] for sh in self.shards], "max")

    def _gather_red(self):
        """every shard's red[0..Q] (Q field sums + max difference) -> all shards, at red[RED_GATHER_OFFSET + r*(Q+1)]: ONE
        collective per sweep; k_finalize folds the rows (sum / max) in rank order, identically on every shard. The
        gathered rows start behind the largest possible input (Q = 16: 17 values), so input and output never overlap."""
        n = self.Q + 1
        w = self.comm.world
        o = RED_GATHER_OFFSET
        assert n <= o
        self.comm.all_gather([sh.red[o:o + w * n] for sh in self.shards], [sh.red[:n] for sh in self.shards])

    def _queue_sweep(self, j):
        """sweep j reads a table whose halo is already in place (shipped during sweep j-1 or by _begin);
        the new marginals of chunk c travel while chunk c+1 is swept"""
        works = []
        fused = self.fused
        ev = self._phase_events() if self._timing else None
        if ev:
            ev[0].record()
        for c in range(self.plans[0].n_chunks):
            for sh in self.shards:
                sh.sweep_chunk(j, c)
            works += self._exchange_chunk(j + 1, c, packed=fused)
        if ev:
            ev[1].record()
        if not fused:
            for w in works:
                w.wait()
            works = []
        for sh in self.shards:
            if not fused:
                sh.unpack(j + 1)
            sh.sweep_fold()  # local folds overlap with the last chunk's exchange (they do not touch the halo)
        self._gather_red()
        for sh in self.shards:
            sh.finalize(0, self.comm.world)
        if ev:
            ev[2].record()
        for w in works:  # the next sweep reads the receive buffers: its kernels wait for the exchanges here
            w.wait()
        if ev:
            ev[3].record()

    # -- where a sweep's time goes on this rank's stream (bench diagnostics, only while timing is on) ------------------
    def _phase_events(self):
        import torch
        quad = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self._phase_log.append(quad)
        return quad

    def phase_times(self):
        """mean ms per sweep on the compute stream: the chunk kernels (with whatever exchange time they could not hide
        behind), fold + field all-gather + finalize, and the wait for the exchanges still in flight after that"""
        import torch
        if not self._phase_log:
            return None
        torch.cuda.synchronize()
        n = len(self._phase_log)
        out = {"chunks_ms": sum(q[0].elapsed_time(q[1]) for q in self._phase_log) / n,
               "reduce_ms": sum(q[1].elapsed_time(q[2]) for q in self._phase_log) / n,
               "exchange_wait_ms": sum(q[2].elapsed_time(q[3]) for q in self._phase_log) / n, "sweeps": n}
        self._phase_log = []
        return out

    def _begin(self, armed):
        works = []
        for c in range(self.plans[0].n_chunks):  # halo of the table the first sweep reads
            works += self._exchange_chunk(0, c)
        for w in works:
            w.wait()
        for sh in self.shards:
            sh.unpack(0)
            sh.begin(armed)
            sh.field_partial(0)
        self._gather_red()
        for sh in self.shards:
            sh.finalize(1, self.comm.world)

    def _exact_diff(self):
        for sh in self.shards:
            sh.msgdiff_partial()
        self._reduce(0, 1)
        for sh in self.shards:
            sh.sync()
        return float(self.shards[0].red[0].item())

    def _run(self, crit, max_sweeps, check_every, want_diff):
        """the convergence decision runs on the device (2-step hints arm the exact 1-step criterion, which sets the stop
        flag: kernels.h dev_params); identical on every shard because k_finalize folds the same gathered rows"""
        self._begin(crit if crit > 0 else -1.0)
        done, st = 0, None
        while done < max_sweeps:
            batch = min(check_every, max_sweeps - done)
            for b in range(batch):
                self._queue_sweep(done + b)
            states = [sh.poll() for sh in self.shards]
            st = states[0]
            done += batch
            if st.stop:
                break
        executed = st.sweep_idx if st is not None else 0
        for sh in self.shards:
            sh.commit(executed)
        niter = st.conv_iter if st is not None else -1
        exact = st.maxdiff if st is not None else None
        if executed and want_diff and not st.last_exact:  # the last sweep reported a 2-step hint
            exact = self._exact_diff()
        self.total_sweeps += executed
        return niter, exact

    def sweep(self, n_sweeps=1, dumping_rate=1.0, want_diff=True):
        if dumping_rate != 1.0:
            raise NotImplementedError("sharded engines run the marginal-gather sweep: damping must be 1")
        _, exact = self._run(-1.0, n_sweeps, max(64, n_sweeps), want_diff)
        return exact

    def converge(self, conv_crit, time_conv, dumping_rate=1.0, check_every=4):
        if dumping_rate != 1.0:
            raise NotImplementedError("sharded engines run the marginal-gather sweep: damping must be 1")
        return self._run(conv_crit, time_conv, check_every, True)

    # -- reductions over the marginals ------------------------------------------------------------
    def _row_sums(self):
        T = 2 * self.Q + self.Q * self.Q
        for sh in self.shards:
            sh.rowsums_partial()
        self._reduce(T, 0)
        for sh in self.shards:
            sh.sync()
        return self.shards[0].red[:T].cpu().numpy().copy()

    def compute_overlap(self):
        """compute_overlap (belief_propagation.cpp:775-811) from the all-reduced confusion matrix"""
        import itertools
        Q = self.Q
        Cm = self._row_sums()[2 * Q:].reshape(Q, Q)
        if Q > 8:  # the reference scores the identity labelling only (belief_propagation.cpp:784-790)
            return float(np.trace(Cm)) / self.N_global
        return max(sum(Cm[a, p[a]] for a in range(Q)) for p in itertools.permutations(range(Q))) / self.N_global

    def na_expect(self):
        return self._row_sums()[:self.Q]

    # -- reductions that need messages: free energy, entropy, EM expectations -------------------------
    def _refresh(self):
        """halo of the current marginal table and the exact global field (h from the current marginals)"""
        works = []
        for c in range(self.plans[0].n_chunks):
            works += self._exchange_chunk(0, c)
        for w in works:
            w.wait()
        for sh in self.shards:
            sh.unpack(0)
            sh.field_partial(0)
        self._gather_red()
        for sh in self.shards:
            sh.finalize(1, self.comm.world)

    def _free_energy_and_entropy(self, want_entropy):
        self._refresh()
        for sh in self.shards:
            sh.fe_partial(want_entropy)
        self._reduce(5, 0)
        fe = [sh.fe_finish() for sh in self.shards][0]
        if self.dc == 0 and self.N_global <= EXACT_NONEDGE_MAX and all(hasattr(sh, "nonedge_exact_partial") for sh in self.shards):
            # small graphs: the O(N^2) loop of the reference, exactly (as the single engine does): every shard gets the
            # marginals of all vertices (own rows all-reduced into a global table) and sums its own rows against them
            tabs = [sh.own_rows_in_global_table(self.N_global) for sh in self.shards]
            self.comm.all_reduce(tabs, "sum")
            for sh, t in zip(self.shards, tabs):
                sh.nonedge_exact_partial(t, want_entropy)
            self._reduce(4, 0)
            for sh in self.shards:
                sh.sync()
            v = self.shards[0].red[:4].cpu().numpy()
            two_n = 2.0 * self.N_global
            return fe, np.array([(v[0] - v[2]) / two_n, (v[1] - v[3]) / two_n])
        nk = [sh.nonedge_partial(want_entropy) for sh in self.shards]
        n, order = nk[0]
        if n:
            self._reduce(n, 0)
        ne = [sh.nonedge_finish(want_entropy, order) for sh in self.shards][0]
        return fe, ne

    def compute_free_energy(self, parts=False):
        """compute_free_energy (belief_propagation.cpp:744-750) over all shards"""
        fe, ne = self._free_energy_and_entropy(False)
        p = np.array([fe[0], fe[1], ne[0]])
        f = -p[0] + p[1] + p[2]
        return (f, p) if parts else f

    def compute_entropy(self, parts=False):
        """compute_entropy (belief_propagation.cpp:752-758); NaN for deg_corr_flag != 0 as the reference"""
        if self.dc != 0:
            nan = float("nan")
            return (nan, np.array([nan, nan, 0.0])) if parts else nan
        fe, ne = self._free_energy_and_entropy(True)
        p = np.array([fe[2], fe[3], ne[1]])
        e = -p[0] + p[1] - p[2]
        return (e, p) if parts else e

    def em_expectations(self):
        """compute_na_expect + compute_cab_expect (belief_propagation.cpp:428-440, 892-989)"""
        self._refresh()
        n = [sh.em_partial() for sh in self.shards][0]
        self._reduce(n, 0)
        return [sh.em_finish() for sh in self.shards][0]

    def inference(self, conv_crit, time_conv, dumping_rate=1.0):
        """belief_propagation::inference (belief_propagation.cpp:77-99)"""
        niter, last = self.converge(conv_crit, time_conv, dumping_rate)
        return dict(niter=niter, last_maxdiff=last, free_energy=self.compute_free_energy(), entropy=self.compute_entropy(),
                    overlap=self.compute_overlap())

    def learning(self, learning_conv_crit, learning_max_time, learning_rate, dumping_rate=1.0):
        """belief_propagation::learning + learning_step (belief_propagation.cpp:14-75) over shards: the same
        stopping rule (float criterion shrinking by 0.1, B3) and integer truncation of na (B8). A parameter
        change leaves the shards' (psi, m) pair slightly inconsistent; the next converge absorbs it."""
        crit = np.float32(learning_conv_crit)
        lr = float(np.float32(learning_rate))
        fold, fdiff, steps, status, sweeps0 = 0.0, 1.0, 0, 0, self.total_sweeps
        N, Q = self.N_global, self.Q
        for sh in self.shards:  # field relaxation inside the EM loop (sbmbp_set_learning_schedule)
            sh.set_schedule(LEARN_FIELD_MIX)
        try:
            return self._learning_loop(crit, lr, fold, fdiff, steps, status, sweeps0, N, Q, learning_max_time, dumping_rate)
        finally:
            for sh in self.shards:
                sh.set_schedule(1.0)

    def _learning_loop(self, crit, lr, fold, fdiff, steps, status, sweeps0, N, Q, learning_max_time, dumping_rate):
        for _ in range(int(learning_max_time)):
            if fdiff < float(crit):
                crit = np.float32(float(crit) * 0.1)
            self.converge(float(crit), int(learning_max_time), dumping_rate)
            na_e, nna_e, cab_e = self.em_expectations()
            fnew = self.compute_free_energy()
            fdiff, fold = abs(fnew - fold), fnew
            if not np.isfinite(fold):
                status = 2
                break
            if fdiff < float(crit):
                status = 1
                break
            na = self.na.astype(np.int64)
            rest = N
            snap = min(LEARN_SNAP * N * float(crit), 0.01)
            for i in range(Q - 1):
                na[i] = int(lr * na_e[i] + (1.0 - lr) * na[i] + snap)
                rest -= na[i]
            na[Q - 1] = rest
            cab = lr * cab_e + (1.0 - lr) * self.cab
            self.expand_bp_params(cab, na.astype(np.uint32), self.beta)
            steps += 1
        return dict(em_steps=steps, status=status, free_energy=fold, overlap=self.compute_overlap(),
                    total_sweeps=self.total_sweeps - sweeps0, cab=self.cab.copy(), na=self.na.copy())

    def local_state(self):
        return [sh.get_state() for sh in self.shards]

    # -- bench plumbing ---------------------------------------------------------------------------
    def stats(self):
        s = self.shards[0].stats()
        return s

    def reset_stats(self):
        for sh in self.shards:
            sh.reset_stats()

    def set_timing(self, on):
        self._timing = bool(on) and self.shards and hasattr(self.shards[0], "torch")  # HIP shards only
        self._phase_log = []
        for sh in self.shards:
            sh.set_timing(on)
