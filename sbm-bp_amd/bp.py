"""Host-side mirror of the reference's BP interface over the C ABI (include/sbmbp.h).

Names follow the reference (belief_propagation.h:96-142, blockmodel.h:18-107,
graph_utilities.h:14-22). The std::mt19937 engine object of the reference cannot cross the ABI;
`seed` takes its place (the engine draws the same stream from it).
"""
import ctypes as C
import os

import numpy as np

from sbm_bp_amd.capi import (InferResult, LearnResult, Stats, c_dp, c_i32p, c_u32p, c_u64p, check, load_library)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


class Graph:
    """Flat CSR adjacency in the engine's layout (replaces adj_list_t, types.h:14-15)."""

    def __init__(self, handle):
        self._lib = load_library()
        self._h = C.c_void_p(handle)
        self.N = self._lib.sbmbp_graph_num_vertices(self._h)
        self.E2 = self._lib.sbmbp_graph_num_directed_edges(self._h)
        self.max_degree = self._lib.sbmbp_graph_max_degree(self._h)

    @classmethod
    def from_edges(cls, pairs, num_vertices=0):
        """edge_to_adj (graph_utilities.cpp:60-77)"""
        lib = load_library()
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        h = C.c_void_p()
        check(lib.sbmbp_graph_from_edges(C.byref(h), pairs.ctypes.data_as(c_u32p), pairs.shape[0], num_vertices))
        return cls(h.value)

    @classmethod
    def from_csr(cls, row_ptr, nbr, rev=None):
        lib = load_library()
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        nbr = np.ascontiguousarray(nbr, dtype=np.uint32)
        rv = None if rev is None else np.ascontiguousarray(rev, dtype=np.uint32)
        h = C.c_void_p()
        check(lib.sbmbp_graph_from_csr(C.byref(h), len(row_ptr) - 1, len(nbr), row_ptr.ctypes.data_as(c_u64p),
                                       nbr.ctypes.data_as(c_u32p), None if rv is None else rv.ctypes.data_as(c_u32p)))
        return cls(h.value)

    def csr(self):
        row_ptr = np.zeros(self.N + 1, dtype=np.uint64)
        nbr = np.zeros(self.E2, dtype=np.uint32)
        rev = np.zeros(self.E2, dtype=np.uint32)
        check(self._lib.sbmbp_graph_copy_csr(self._h, row_ptr.ctypes.data_as(c_u64p), nbr.ctypes.data_as(c_u32p),
                                             rev.ctypes.data_as(c_u32p)))
        return row_ptr, nbr, rev

    def initial_state(self, Q, flag=0, conf=None, seed=0):
        """the state init_messages (belief_propagation.cpp:101-217) produces from std::mt19937(seed), computed on the
        host without an engine: (psi N x Q, msg_out E2 x Q)"""
        psi = np.empty((self.N, Q), dtype=np.float64)
        msg = np.empty((self.E2, Q), dtype=np.float64)
        cf = None if conf is None else np.ascontiguousarray(conf, dtype=np.int32)
        check(self._lib.sbmbp_host_init_state(self._h, Q, flag, None if cf is None else cf.ctypes.data_as(c_i32p), seed,
                                              psi.ctypes.data_as(c_dp), msg.ctypes.data_as(c_dp)))
        return psi, msg

    def __del__(self):
        try:
            self._lib.sbmbp_graph_destroy(self._h)
        except Exception:
            pass


def load_edge_list(edge_list_path, num_vertices=0):
    """load_edge_list + edge_to_adj (graph_utilities.cpp:42-77). A missing file raises."""
    lib = load_library()
    h = C.c_void_p()
    check(lib.sbmbp_graph_load_edgelist(C.byref(h), os.fsencode(edge_list_path), num_vertices))
    return Graph(h.value)


def load_beliefs(path):
    """load_beliefs (graph_utilities.cpp:8-23): one int per line, -1 = unknown"""
    return np.loadtxt(path, dtype=np.int32, ndmin=1)


def load_confs(path):
    """load_confs (graph_utilities.cpp:25-40)"""
    return np.loadtxt(path, dtype=np.uint32, ndmin=1)


class blockmodel_t:
    """What BP consumes of blockmodel_t (blockmodel.cpp:7-49, getters :51,:89-101)."""

    def __init__(self, graph, Q, deg_corr_flag=0):
        self.graph, self._Q, self._dc = graph, int(Q), int(deg_corr_flag)

    def get_N(self):
        return self.graph.N

    def get_Q(self):
        return self._Q

    def get_E(self):
        return self.graph.E2 // 2

    def get_deg_corr_flag(self):
        return self._dc

    def get_graph_max_degree(self):
        return self.graph.max_degree


class bp_blockmodel_state:
    """types.h:23-26"""

    def __init__(self, cab, na):
        self.cab = np.ascontiguousarray(cab, dtype=np.float64)
        self.na = np.ascontiguousarray(na, dtype=np.uint32)


def bp_param_from_epsilon_c(blockmodel, epsilon, c):
    """blockmodel.cpp:229-272"""
    lib = load_library()
    Q = blockmodel.get_Q()
    cab = np.zeros((Q, Q))
    na = np.zeros(Q, dtype=np.uint32)
    check(lib.sbmbp_param_from_epsilon_c(blockmodel.get_N(), Q, epsilon, c, _dp(cab), na.ctypes.data_as(c_u32p)))
    return bp_blockmodel_state(cab, na)


def bp_param_from_direct(blockmodel, pa, cab):
    """blockmodel.cpp:274-302 (cab = upper triangle, row-major)"""
    lib = load_library()
    Q = blockmodel.get_Q()
    pa = np.ascontiguousarray(pa, dtype=np.float64)
    cu = np.ascontiguousarray(cab, dtype=np.float64)
    if len(pa) != Q or len(cu) != Q * (Q + 1) // 2:
        raise ValueError("pa needs Q entries and cab Q(Q+1)/2 entries")
    full = np.zeros((Q, Q))
    na = np.zeros(Q, dtype=np.uint32)
    check(lib.sbmbp_param_from_direct(blockmodel.get_N(), Q, _dp(pa), _dp(cu), _dp(full), na.ctypes.data_as(c_u32p)))
    return bp_blockmodel_state(full, na)


class BeliefPropagation:
    """class belief_propagation (belief_propagation.h:18-142) on the GPU engine."""

    conditional = True

    def __init__(self, device=-1):
        self._lib = load_library()
        self._h = None
        self._device = device
        self._beta = 1.0
        self.if_output_marginals_ = False
        self.Q = self.N = self.E2 = 0

    def __del__(self):
        try:
            if self._h:
                self._lib.sbmbp_destroy(self._h)
        except Exception:
            pass

    # -- life cycle -------------------------------------------------------------------------
    def bp_allocate(self, blockmodel):
        """belief_propagation.cpp:223-288"""
        if self._h:
            self._lib.sbmbp_destroy(self._h)
            self._h = None
        h = C.c_void_p()
        check(self._lib.sbmbp_create(C.byref(h), blockmodel.graph._h, blockmodel.get_Q(), blockmodel.get_deg_corr_flag(),
                                     self._device))
        self._h = h
        self._graph = blockmodel.graph  # the reference keeps a raw pointer (belief_propagation.h:27)
        self.Q, self.N, self.E2 = blockmodel.get_Q(), blockmodel.get_N(), blockmodel.graph.E2

    def init_messages(self, blockmodel, bp_messages_init_flag, conf, true_conf, seed):
        """belief_propagation.cpp:101-217"""
        self.bp_allocate(blockmodel)
        tc = np.ascontiguousarray(true_conf, dtype=np.uint32)
        if len(tc) != self.N:
            raise ValueError("true_conf needs N entries")
        cf = None
        if conf is not None and len(conf):
            cf = np.ascontiguousarray(conf, dtype=np.int32)
            if len(cf) != self.N:
                raise ValueError("conf needs N entries (-1 = unknown)")
        check(self._lib.sbmbp_init_messages(self._h, bp_messages_init_flag, None if cf is None else cf.ctypes.data_as(c_i32p),
                                            tc.ctypes.data_as(c_u32p), seed, int(self.conditional)))

    def init_messages_device(self, blockmodel, true_conf, seed):
        self.bp_allocate(blockmodel)
        tc = np.ascontiguousarray(true_conf, dtype=np.uint32)
        check(self._lib.sbmbp_init_messages_device(self._h, seed, tc.ctypes.data_as(c_u32p)))

    def reinit_messages_device(self, true_conf, seed):
        """the same device-side initial state again on the existing engine (no reallocation)"""
        tc = np.ascontiguousarray(true_conf, dtype=np.uint32)
        check(self._lib.sbmbp_init_messages_device(self._h, seed, tc.ctypes.data_as(c_u32p)))

    def init_special_needs(self, if_output_marginals):
        self.if_output_marginals_ = bool(if_output_marginals)

    def set_beta(self, beta):
        self._beta = float(beta)

    def expand_bp_params(self, state):
        """belief_propagation.cpp:290-317"""
        cab = np.ascontiguousarray(state.cab, dtype=np.float64)
        na = np.ascontiguousarray(state.na, dtype=np.uint32)
        check(self._lib.sbmbp_set_params(self._h, _dp(cab), na.ctypes.data_as(c_u32p), self._beta))

    def set_schedule(self, field_mix=1.0, check_every=1):
        check(self._lib.sbmbp_set_schedule(self._h, field_mix, check_every))

    def set_gather_mode(self, mode=0):
        """0 = automatic (marginal-gather sweep when exact), 1 = always gather messages"""
        check(self._lib.sbmbp_set_gather_mode(self._h, mode))

    def set_auto_relax(self, on=True):
        """adaptive relaxation of converge / inference / learning (sbmbp.h); off = plain synchronous sweeps"""
        check(self._lib.sbmbp_set_auto_relax(self._h, int(on)))

    def relaxation(self):
        """(field level, generic level, field_mix, damping factor) the last converge call ended on; (0, -1, ., 1.0): never relaxed"""
        fl, gl, mix, dmp = C.c_int(0), C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        check(self._lib.sbmbp_get_relaxation(self._h, C.byref(fl), C.byref(gl), C.byref(mix), C.byref(dmp)))
        return fl.value, gl.value, mix.value, dmp.value

    def set_learning_schedule(self, field_mix=0.3, snap=1.0):
        """field relaxation inside the EM loop's BP runs and the snap tolerance of the group-size truncation (sbmbp.h)"""
        check(self._lib.sbmbp_set_learning_schedule(self._h, field_mix, snap))

    def set_nonedge_mode(self, mode=0, series_order=0):
        check(self._lib.sbmbp_set_nonedge_mode(self._h, mode, series_order))

    # -- hot path ---------------------------------------------------------------------------
    def converge(self, conv_crit, time_conv, dumping_rate):
        """belief_propagation.cpp:386-415; returns (niter, last max|delta|)"""
        niter, last = C.c_int(0), C.c_double(0.0)
        check(self._lib.sbmbp_converge(self._h, conv_crit, time_conv, dumping_rate, C.byref(niter), C.byref(last)))
        return niter.value, last.value

    def sweep(self, n_sweeps=1, dumping_rate=1.0, want_diff=True):
        """exactly n_sweeps synchronous sweeps; returns the last max|delta message| (an extra streaming
        pass over both message buffers in the marginal-gather form) unless want_diff is False"""
        if not want_diff:
            check(self._lib.sbmbp_sweep(self._h, dumping_rate, n_sweeps, None))
            return None
        last = C.c_double(0.0)
        check(self._lib.sbmbp_sweep(self._h, dumping_rate, n_sweeps, C.byref(last)))
        return last.value

    def compute_free_energy(self, parts=False):
        f = C.c_double(0.0)
        p = np.zeros(3)
        check(self._lib.sbmbp_free_energy(self._h, C.byref(f), _dp(p)))
        return (f.value, p) if parts else f.value

    def compute_entropy(self, parts=False):
        e = C.c_double(0.0)
        p = np.zeros(3)
        check(self._lib.sbmbp_entropy(self._h, C.byref(e), _dp(p)))
        return (e.value, p) if parts else e.value

    def compute_overlap(self):
        ov = C.c_double(0.0)
        check(self._lib.sbmbp_overlap(self._h, C.byref(ov)))
        return ov.value

    def confusion(self):
        Cm = np.zeros((self.Q, self.Q))
        check(self._lib.sbmbp_confusion(self._h, _dp(Cm)))
        return Cm

    def em_expectations(self, cab=True):
        """compute_na_expect + compute_cab_expect (belief_propagation.cpp:428-440, 892-989); cab=False: the group sizes only
        (above Q = 16 the cab expectations, i.e. -m learn, are not implemented)"""
        na, nna, cabe = np.zeros(self.Q), np.zeros(self.Q), np.zeros((self.Q, self.Q))
        check(self._lib.sbmbp_em_expectations(self._h, _dp(na), _dp(nna), _dp(cabe) if cab else None))
        return na, nna, cabe

    def inference(self, blockmodel, state, conv_crit, time_conv, dumping_rate):
        """belief_propagation.cpp:77-99; returns the result struct (format_infer_line prints it)"""
        self.expand_bp_params(state)
        res = InferResult()
        check(self._lib.sbmbp_inference(self._h, conv_crit, time_conv, dumping_rate, C.byref(res)))
        return res

    def learning(self, blockmodel, state, learning_conv_crit, learning_max_time, learning_rate, dumping_rate):
        """belief_propagation.cpp:14-51"""
        self.expand_bp_params(state)
        res = LearnResult()
        check(self._lib.sbmbp_learning(self._h, learning_conv_crit, learning_max_time, learning_rate, dumping_rate,
                                       C.byref(res)))
        return res

    # -- state access -----------------------------------------------------------------------
    def get_params(self):
        cab = np.zeros((self.Q, self.Q))
        na = np.zeros(self.Q, dtype=np.uint32)
        check(self._lib.sbmbp_get_params(self._h, _dp(cab), na.ctypes.data_as(c_u32p)))
        return cab, na

    def get_state(self, psi=True, msg=True):
        p = np.zeros((self.N, self.Q)) if psi else None
        m = np.zeros((self.E2, self.Q)) if msg else None
        check(self._lib.sbmbp_get_state(self._h, _dp(p), _dp(m)))
        return p, m

    def set_state(self, psi, msg_out):
        p = None if psi is None else np.ascontiguousarray(psi, dtype=np.float64)
        m = None if msg_out is None else np.ascontiguousarray(msg_out, dtype=np.float64)
        check(self._lib.sbmbp_set_state(self._h, _dp(p), _dp(m)))

    def real_psi(self):
        return self.get_state(True, False)[0]

    def h(self):
        h = np.zeros(self.Q)
        check(self._lib.sbmbp_get_field(self._h, _dp(h)))
        return h

    def stats(self):
        s = Stats()
        check(self._lib.sbmbp_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        check(self._lib.sbmbp_reset_stats(self._h))

    def set_timing(self, on):
        check(self._lib.sbmbp_set_timing(self._h, int(on)))

    def set_stream(self, hip_stream):
        check(self._lib.sbmbp_set_stream(self._h, C.c_void_p(hip_stream)))


class bp_basic(BeliefPropagation):
    """belief_propagation.cpp:1079-1098 — every row is updated (learn mode, main.cpp:319-320)"""
    conditional = False


class bp_conditional(BeliefPropagation):
    """belief_propagation.cpp:1100-1126 — rows with a planted label are clamped (infer mode)"""
    conditional = True


def format_infer_line(res):
    """the stdout line of belief_propagation.cpp:88 at the default 6 significant digits"""
    def g(x):
        return "-nan" if x != x else "%g" % x
    return "%s %s %s %d \n" % (g(res.entropy), g(res.free_energy), g(res.overlap), res.niter)
