"""TEST INFRASTRUCTURE — a numpy model of the multi-rank protocol of csrc/dist.hip (marginal-gather sweeps: chunked halo
exchange in receive order, one all-gather of Q+1 values per sweep, device-style convergence state), so that the C++ shard
PLAN can be exercised across real processes on the CPU (gloo, world_size 2) without a GPU. The shard kernel is the numpy
stand-in of tests/shard_numpy_backend.py. Never imported by the product."""
import ctypes as C

import numpy as np

RED_GATHER_OFFSET = 32


class CppPlan:
    """the plan of one rank as csrc/dist.hip builds it (sbmbp_plan_summary + sbmbp_plan_arrays), with the attribute names of
    tests/plan_model.ShardPlan"""

    def __init__(self, graph, world, rank, n_chunks):
        from sbm_bp_amd.capi import DistInfo, c_u32p, c_u64p, check, load_library
        lib = load_library()
        info = DistInfo()
        W, Cn = world, max(1, n_chunks)
        sc, rc, mc = (np.zeros(W, dtype=np.uint64) for _ in range(3))
        check(lib.sbmbp_plan_summary(graph._h, world, rank, Cn, C.byref(info), sc.ctypes.data_as(c_u64p), rc.ctypes.data_as(c_u64p),
                                     mc.ctypes.data_as(c_u64p)))
        self.rank, self.world, self.n_chunks = rank, world, Cn
        self.n_global, self.row0, self.n_own, self.n_halo = info.n_global, info.row0, info.n_own, info.n_halo
        self.n_edges, self.n_halo_msgs = int(info.n_edges), int(info.n_halo_msgs)
        self.send_counts, self.recv_counts, self.msg_counts = sc.astype(np.int64), rc.astype(np.int64), mc.astype(np.int64)
        n_send = int(sc.sum())
        self.nbr_local = np.zeros(self.n_edges, dtype=np.uint32)
        self.halo_global = np.zeros(self.n_halo, dtype=np.uint32)
        self.chunk_row = np.zeros(Cn + 1, dtype=np.uint32)
        scp, rcp = np.zeros((Cn, W), dtype=np.uint64), np.zeros((Cn, W), dtype=np.uint64)
        self.send_idx_chunked = np.zeros(n_send, dtype=np.uint32)
        self.snd_ptr = np.zeros(self.n_own + 1, dtype=np.uint32)
        self.snd_slot = np.zeros(n_send, dtype=np.uint32)
        self.rev_local = np.zeros(self.n_edges, dtype=np.uint32)
        self.msg_send_edge = np.zeros(self.n_halo_msgs, dtype=np.uint32)
        p32 = lambda a: a.ctypes.data_as(c_u32p)  # noqa: E731
        check(lib.sbmbp_plan_arrays(graph._h, world, rank, Cn, p32(self.nbr_local), p32(self.halo_global), p32(self.chunk_row),
                                    scp.ctypes.data_as(c_u64p), rcp.ctypes.data_as(c_u64p), p32(self.send_idx_chunked), p32(self.snd_ptr),
                                    p32(self.snd_slot), p32(self.rev_local), p32(self.msg_send_edge)))
        self.send_counts_cp, self.recv_counts_cp = scp.astype(np.int64), rcp.astype(np.int64)
        self.send_idx_chunked = self.send_idx_chunked.astype(np.int64)
        row_ptr = graph.csr()[0].astype(np.int64)
        self.edge0 = int(row_ptr[self.row0])
        self.row_ptr = (row_ptr[self.row0:self.row0 + self.n_own + 1] - self.edge0).astype(np.uint64)
        self.deg = np.diff(self.row_ptr.astype(np.int64))
        self.send_off_cp = np.concatenate([[0], np.cumsum(self.send_counts_cp.ravel())])[:-1].reshape(Cn, W)
        self.send_off_c = np.concatenate([[0], np.cumsum(self.send_counts_cp.sum(1))])
        self.stage_off_c = np.concatenate([[0], np.cumsum(self.recv_counts_cp.sum(1))])
        self.stage_to_halo = np.arange(self.n_halo, dtype=np.int64)


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



class ProtocolModel:
    """the sweep / converge / overlap protocol over vertex-range shards, one or several ranks per process"""

    def __init__(self, plans, Q, dc, comm, backend_factory=None):
        self.comm, self.Q, self.dc = comm, Q, dc
        self.plans = plans
        self.shards = [backend_factory(p) for p in plans]
        self.fused = all(getattr(sh, "fused", False) for sh in self.shards)
        self._timing, self._phase_log = False, []
        self.N_global = plans[0].n_global
        self.E2_local = sum(p.n_edges for p in plans)
        self.E2_global = None
        self.total_sweeps = 0

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
        ev = None
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

