"""Multi-GPU BP: Python face of the C++ driver (csrc/dist.hip, include/sbmbp.h section "Multi-GPU").

One rank = one GPU = one `ShardedBP`. The plan (vertex ranges, halo, send lists, cut edges), the communicator
(RCCL over xGMI) and the sweep / reduction loops all live behind the C ABI; Python only boots the ranks:

  * `Comm.rccl_from_torch()`   — one process per GPU (bench.py, torchrun): rank 0 draws the RCCL id, torch.distributed
                                  hands it to the others, every later collective is RCCL inside the library;
  * `Comm.callbacks_from_torch()` — rehearsal on a one-GPU box: the same driver with gloo between the processes (buffers
                                  staged through the host by the library);
  * `LocalShards`               — the ranks are threads of this process sharing one GPU (tests, budget measurements).

No reference counterpart: junipertcy/sbm-bp is single-process.
"""
import ctypes as C
from concurrent.futures import FIRST_EXCEPTION, ThreadPoolExecutor, wait

import numpy as np

from sbm_bp_amd.capi import (COMM_ID_BYTES, CommCallbacks, DistInfo, InferResult, LearnResult, Stats, c_dp, c_i32p, c_u32p, c_u64p,
                             check, load_library)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


class Comm:
    """sbmbp_comm_t"""

    def __init__(self, handle, keep=None):
        self._lib = load_library()
        self._h = C.c_void_p(handle)
        self._keep = keep  # callback objects must outlive the communicator
        self.rank = self._lib.sbmbp_comm_rank(self._h)
        self.world = self._lib.sbmbp_comm_size(self._h)
        self.transport = self._lib.sbmbp_comm_transport(self._h).decode()

    def __del__(self):
        try:
            self._lib.sbmbp_comm_destroy(self._h)
        except Exception:
            pass

    def abort(self):
        self._lib.sbmbp_comm_abort(self._h)

    @classmethod
    def rccl_from_torch(cls, device):
        """the default process group only carries the id: rank 0 draws it, everybody joins the RCCL communicators"""
        import torch
        import torch.distributed as dist
        lib = load_library()
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        if rank == 0:
            check(lib.sbmbp_comm_unique_id(buf))
        dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0)
        ident = bytes(t.cpu().tolist())
        h = C.c_void_p()
        check(lib.sbmbp_comm_init_rank(C.byref(h), ident, world, rank, device))
        return cls(h.value)

    @classmethod
    def local(cls, world):
        lib = load_library()
        arr = (C.c_void_p * world)()
        check(lib.sbmbp_comm_init_local(arr, world))
        return [cls(arr[r]) for r in range(world)]

    @classmethod
    def callbacks_from_torch(cls, group=None):
        """collectives of a gloo process group (default: the default group) on host buffers the library hands over"""
        import torch
        import torch.distributed as dist
        lib = load_library()
        rank, world = dist.get_rank(), dist.get_world_size()

        def view(ptr, n):
            n = int(n)
            return torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(max(n, 1),)))[:n]

        def exchange(user, send, send_rows, recv, recv_rows, width):
            try:
                sc = [int(send_rows[p]) * width for p in range(world)]
                rc = [int(recv_rows[p]) * width for p in range(world)]
                dist.all_to_all_single(view(recv, sum(rc)), view(send, sum(sc)), rc, sc, group=group)
                return 0
            except Exception as ex:  # never let an exception cross the C boundary
                print("exchange callback:", ex, flush=True)
                return 1

        def allgather(user, inp, n, out):
            try:
                parts = [torch.empty(int(n), dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, view(inp, n).clone(), group=group)
                view(out, n * world).copy_(torch.cat(parts))
                return 0
            except Exception as ex:
                print("allgather callback:", ex, flush=True)
                return 1

        def allreduce(user, buf, n, op):
            try:
                dist.all_reduce(view(buf, n), op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
                return 0
            except Exception as ex:
                print("allreduce callback:", ex, flush=True)
                return 1

        cb = CommCallbacks(None, CommCallbacks.EXCHANGE(exchange), CommCallbacks.ALLGATHER(allgather), CommCallbacks.ALLREDUCE(allreduce))
        h = C.c_void_p()
        check(lib.sbmbp_comm_init_callbacks(C.byref(h), world, rank, C.byref(cb)))
        return cls(h.value, keep=cb)


class ShardedBP:
    """one rank of belief_propagation over vertex-range shards (sbmbp_dist_t); same method names as BeliefPropagation"""

    def __init__(self, graph, Q, dc, comm, device=0, n_chunks=0):
        self._lib = load_library()
        self.comm, self.graph, self.Q, self.dc = comm, graph, int(Q), int(dc)
        self._h = None
        h = C.c_void_p()
        check(self._lib.sbmbp_dist_create(C.byref(h), comm._h, graph._h, Q, dc, device, n_chunks))
        self._h = h
        info = DistInfo()
        check(self._lib.sbmbp_dist_info(self._h, C.byref(info)))
        self.info = info
        self.N_global, self.E2_global = info.n_global, info.e2_global
        self.row0, self.n_own, self.n_halo, self.n_edges = info.row0, info.n_own, info.n_halo, info.n_edges
        self.cab = self.na = None
        self.beta = 1.0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if self._h:
            self._lib.sbmbp_dist_destroy(self._h)
            self._h = None

    # -- construction -------------------------------------------------------------------------
    @classmethod
    def synthetic(cls, N, Q, c, eps, graph_seed, dc=0, seed=1234, comm=None, device=0, n_chunks=0, graph=None):
        """every rank generates the same planted-partition graph and keeps its own row range (graph: the host graph of an
        earlier call with the same arguments, kept in .graph, to build another plan on it)"""
        import sbm_bp_amd as S
        from sbm_bp_amd import synth
        cin, cout = synth.cin_cout(Q, c, eps)
        g = graph
        if g is None:
            pairs, cin, cout = synth.planted_partition(N, Q, c, eps, graph_seed)
            g = S.Graph.from_edges(pairs, N)
            del pairs
        self = cls(g, Q, dc, comm, device, n_chunks)
        self.graph = g
        self.init_messages_device(seed, synth.true_conf(N, Q))
        self.expand_bp_params(synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(N, Q), dtype=np.uint32), 1.0)
        return self

    # -- state and parameters -------------------------------------------------------------------
    def init_messages_device(self, seed, true_conf_global):
        tc = np.ascontiguousarray(true_conf_global, dtype=np.uint32)
        check(self._lib.sbmbp_dist_init_messages_device(self._h, seed, tc.ctypes.data_as(c_u32p)))

    def init_messages(self, flag, conf, true_conf, seed, conditional=True):
        """init_messages (belief_propagation.cpp:101-217): the reference's std::mt19937 stream, this rank's slice"""
        tc = np.ascontiguousarray(true_conf, dtype=np.uint32)
        cf = None if conf is None or len(conf) == 0 else np.ascontiguousarray(conf, dtype=np.int32)
        check(self._lib.sbmbp_dist_init_messages(self._h, flag, None if cf is None else cf.ctypes.data_as(c_i32p),
                                                 tc.ctypes.data_as(c_u32p), seed, int(conditional)))

    def set_state(self, psi_own, msg_own):
        p = np.ascontiguousarray(psi_own, dtype=np.float64)
        m = np.ascontiguousarray(msg_own, dtype=np.float64)
        check(self._lib.sbmbp_dist_set_state(self._h, _dp(p), _dp(m)))

    def get_state(self):
        psi = np.zeros((self.n_own, self.Q))
        msg = np.zeros((self.n_edges, self.Q))
        check(self._lib.sbmbp_dist_get_state(self._h, _dp(psi), _dp(msg)))
        return psi, msg

    def gather_marginals(self):
        psi = np.zeros((self.N_global, self.Q))
        check(self._lib.sbmbp_dist_gather_marginals(self._h, _dp(psi)))
        return psi

    def expand_bp_params(self, cab, na, beta=1.0):
        self.cab = np.ascontiguousarray(cab, dtype=np.float64)
        self.na = np.ascontiguousarray(na, dtype=np.uint32)
        self.beta = float(beta)
        check(self._lib.sbmbp_dist_set_params(self._h, _dp(self.cab), self.na.ctypes.data_as(c_u32p), self.beta))

    def get_params(self):
        cab = np.zeros((self.Q, self.Q))
        na = np.zeros(self.Q, dtype=np.uint32)
        check(self._lib.sbmbp_dist_get_params(self._h, _dp(cab), na.ctypes.data_as(c_u32p)))
        return cab, na

    def set_schedule(self, field_mix=1.0, check_every=8):
        check(self._lib.sbmbp_dist_set_schedule(self._h, field_mix, check_every))

    def set_learning_schedule(self, field_mix=0.3, snap=1.0):
        check(self._lib.sbmbp_dist_set_learning_schedule(self._h, field_mix, snap))

    def set_gather_mode(self, mode=0):
        check(self._lib.sbmbp_dist_set_gather_mode(self._h, mode))

    def set_auto_relax(self, on=True):
        check(self._lib.sbmbp_dist_set_auto_relax(self._h, int(on)))

    def relaxation(self):
        fl, gl = C.c_int(0), C.c_int(0)
        check(self._lib.sbmbp_dist_get_relaxation(self._h, C.byref(fl), C.byref(gl)))
        return fl.value, gl.value

    # -- hot path ---------------------------------------------------------------------------------
    def sweep(self, n_sweeps=1, dumping_rate=1.0, want_diff=True):
        if not want_diff:
            check(self._lib.sbmbp_dist_sweep(self._h, dumping_rate, n_sweeps, None))
            return None
        last = C.c_double(0.0)
        check(self._lib.sbmbp_dist_sweep(self._h, dumping_rate, n_sweeps, C.byref(last)))
        return last.value

    def converge(self, conv_crit, time_conv, dumping_rate=1.0, check_every=None):
        if check_every is not None:
            self.set_schedule(1.0, check_every)
        niter, last = C.c_int(0), C.c_double(0.0)
        check(self._lib.sbmbp_dist_converge(self._h, conv_crit, time_conv, dumping_rate, C.byref(niter), C.byref(last)))
        return niter.value, last.value

    # -- reductions ---------------------------------------------------------------------------------
    def compute_free_energy(self, parts=False):
        f, p = C.c_double(0.0), np.zeros(3)
        check(self._lib.sbmbp_dist_free_energy(self._h, C.byref(f), _dp(p)))
        return (f.value, p) if parts else f.value

    def compute_entropy(self, parts=False):
        e, p = C.c_double(0.0), np.zeros(3)
        check(self._lib.sbmbp_dist_entropy(self._h, C.byref(e), _dp(p)))
        return (e.value, p) if parts else e.value

    def em_expectations(self):
        na, nna, cab = np.zeros(self.Q), np.zeros(self.Q), np.zeros((self.Q, self.Q))
        check(self._lib.sbmbp_dist_em_expectations(self._h, _dp(na), _dp(nna), _dp(cab)))
        return na, nna, cab

    def compute_overlap(self):
        ov = C.c_double(0.0)
        check(self._lib.sbmbp_dist_overlap(self._h, C.byref(ov)))
        return ov.value

    def confusion(self):
        Cm = np.zeros((self.Q, self.Q))
        check(self._lib.sbmbp_dist_confusion(self._h, _dp(Cm)))
        return Cm

    def inference(self, conv_crit, time_conv, dumping_rate=1.0):
        res = InferResult()
        check(self._lib.sbmbp_dist_inference(self._h, conv_crit, time_conv, dumping_rate, C.byref(res)))
        return dict(niter=res.niter, last_maxdiff=res.last_maxdiff, free_energy=res.free_energy, entropy=res.entropy, overlap=res.overlap)

    def learning(self, learning_conv_crit, learning_max_time, learning_rate, dumping_rate=1.0):
        res = LearnResult()
        check(self._lib.sbmbp_dist_learning(self._h, learning_conv_crit, learning_max_time, learning_rate, dumping_rate, C.byref(res)))
        cab, na = self.get_params()
        self.cab, self.na = cab, na
        return dict(em_steps=res.em_steps, status=res.status, free_energy=res.free_energy, overlap=res.overlap,
                    total_sweeps=res.total_sweeps, cab=cab.copy(), na=na.copy())

    # -- bench plumbing -----------------------------------------------------------------------------
    def stats(self):
        s = Stats()
        check(self._lib.sbmbp_dist_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        check(self._lib.sbmbp_dist_reset_stats(self._h))

    def set_timing(self, on):
        check(self._lib.sbmbp_dist_set_timing(self._h, int(on)))

    def phase_times(self):
        ms = (C.c_double * 3)()
        n = C.c_uint64(0)
        check(self._lib.sbmbp_dist_phase_times(self._h, ms, C.byref(n)))
        if not n.value:
            return None
        return {"chunks_ms": ms[0], "reduce_ms": ms[1], "exchange_wait_ms": ms[2], "sweeps": n.value}

    def peer_rows(self):
        s = np.zeros(self.comm.world, dtype=np.uint64)
        r = np.zeros(self.comm.world, dtype=np.uint64)
        check(self._lib.sbmbp_dist_peer_rows(self._h, s.ctypes.data_as(c_u64p), r.ctypes.data_as(c_u64p)))
        return s, r


class LocalShards:
    """all ranks of a run as threads of this process (in-process transport): what a multi-GPU run computes, on one GPU.
    Every method runs on all ranks at once (each rank keeps its own thread) and returns rank 0's result."""

    def __init__(self, graph, Q, dc, world, n_chunks=None, device=0):
        self.world, self.graph, self.Q, self.dc = int(world), graph, int(Q), int(dc)
        self._pools = [ThreadPoolExecutor(max_workers=1) for _ in range(self.world)]
        self.comms = Comm.local(self.world)
        nc = 0 if n_chunks is None else int(n_chunks)
        self.ranks = None
        self.ranks = self._all(lambda r: ShardedBP(graph, Q, dc, self.comms[r], device, nc), by_index=True)
        self.N_global, self.E2_global = self.ranks[0].N_global, self.ranks[0].E2_global

    @classmethod
    def from_csr(cls, row_ptr, nbr, Q, dc, world, n_chunks=None, device=0):
        import sbm_bp_amd as S
        return cls(S.Graph.from_csr(row_ptr, nbr), Q, dc, world, n_chunks, device)

    def _all(self, fn, by_index=False):
        futs = [self._pools[r].submit(fn, r if by_index else self.ranks[r]) for r in range(self.world)]
        done, _ = wait(futs, return_when=FIRST_EXCEPTION)
        if any(f.exception() is not None for f in done):  # one rank gave up: wake the others out of their collectives
            for c in self.comms:
                c.abort()
        errs = [f.exception() for f in futs]  # (waits for all)
        # report the rank that failed, not a peer that was woken out of a collective because of it
        first = next((e for e in errs if e is not None and getattr(e, "code", 0) != -8), None) or next((e for e in errs if e is not None), None)
        if first is not None:
            raise first
        return [f.result() for f in futs]

    def close(self):
        if getattr(self, "ranks", None):
            self._all(lambda sh: sh.close())
            self.ranks = None
        for p in getattr(self, "_pools", []):
            p.shutdown(wait=True)
        self._pools = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_messages_device(self, seed, tc):
        self._all(lambda sh: sh.init_messages_device(seed, tc))

    def init_messages(self, flag, conf, true_conf, seed, conditional=True):
        self._all(lambda sh: sh.init_messages(flag, conf, true_conf, seed, conditional))

    def set_state_global(self, psi, msg):
        """psi N x Q and msg E2 x Q of the whole graph: every rank takes its rows"""
        rp = self.graph.csr()[0]

        def put(sh):
            e0 = int(rp[sh.row0])
            sh.set_state(psi[sh.row0:sh.row0 + sh.n_own], msg[e0:e0 + sh.n_edges])
        self._all(put)

    def expand_bp_params(self, cab, na, beta=1.0):
        self._all(lambda sh: sh.expand_bp_params(cab, na, beta))
        self.cab, self.na = self.ranks[0].cab, self.ranks[0].na

    def set_schedule(self, field_mix=1.0, check_every=8):
        self._all(lambda sh: sh.set_schedule(field_mix, check_every))

    def set_gather_mode(self, mode=0):
        self._all(lambda sh: sh.set_gather_mode(mode))

    def set_auto_relax(self, on=True):
        self._all(lambda sh: sh.set_auto_relax(on))

    def relaxation(self):
        return self._all(lambda sh: sh.relaxation())[0]

    def sweep(self, n_sweeps=1, dumping_rate=1.0, want_diff=True):
        return self._all(lambda sh: sh.sweep(n_sweeps, dumping_rate, want_diff))[0]

    def converge(self, conv_crit, time_conv, dumping_rate=1.0, check_every=None):
        out = self._all(lambda sh: sh.converge(conv_crit, time_conv, dumping_rate, check_every))
        assert all(o[0] == out[0][0] for o in out), "ranks disagree on niter"
        return out[0]

    def compute_free_energy(self, parts=False):
        return self._all(lambda sh: sh.compute_free_energy(parts))[0]

    def compute_entropy(self, parts=False):
        return self._all(lambda sh: sh.compute_entropy(parts))[0]

    def em_expectations(self):
        return self._all(lambda sh: sh.em_expectations())[0]

    def compute_overlap(self):
        return self._all(lambda sh: sh.compute_overlap())[0]

    def confusion(self):
        return self._all(lambda sh: sh.confusion())[0]

    def inference(self, conv_crit, time_conv, dumping_rate=1.0):
        return self._all(lambda sh: sh.inference(conv_crit, time_conv, dumping_rate))[0]

    def learning(self, learning_conv_crit, learning_max_time, learning_rate, dumping_rate=1.0):
        out = self._all(lambda sh: sh.learning(learning_conv_crit, learning_max_time, learning_rate, dumping_rate))
        self.cab, self.na = out[0]["cab"], out[0]["na"]
        return out[0]

    def local_state(self):
        return self._all(lambda sh: sh.get_state())

    def global_state(self):
        st = self.local_state()
        return np.concatenate([s[0] for s in st]), np.concatenate([s[1] for s in st])

    def gather_marginals(self):
        return self._all(lambda sh: sh.gather_marginals())[0]

    def stats(self):
        return self._all(lambda sh: sh.stats())

    def reset_stats(self):
        self._all(lambda sh: sh.reset_stats())

    def set_timing(self, on):
        self._all(lambda sh: sh.set_timing(on))
