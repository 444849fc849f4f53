"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (the CPU restatement, bp_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (sbm-bp_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)
c_i32p = C.POINTER(C.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "bp_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        sig = {
            "orc_graph_from_edges": (C.c_void_p, [c_u32p, C.c_uint64, C.c_uint32]),
            "orc_graph_load_edgelist": (C.c_void_p, [C.c_char_p, C.c_uint32]),
            "orc_graph_free": (None, [C.c_void_p]),
            "orc_graph_n": (C.c_uint32, [C.c_void_p]),
            "orc_graph_e2": (C.c_uint64, [C.c_void_p]),
            "orc_graph_copy": (None, [C.c_void_p, c_u64p, c_u32p, c_u32p]),
            "orc_rng_create": (C.c_void_p, [C.c_uint]),
            "orc_rng_free": (None, [C.c_void_p]),
            "orc_rng_draw": (C.c_double, [C.c_void_p]),
            "orc_param_from_epsilon_c": (None, [C.c_uint32, C.c_uint32, C.c_double, C.c_double, c_dp, c_u32p]),
            "orc_param_from_direct": (None, [C.c_uint32, C.c_uint32, c_dp, c_dp, c_dp, c_u32p]),
            "orc_bp_create": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_uint32]),
            "orc_bp_free": (None, [C.c_void_p]),
            "orc_bp_init_messages": (None, [C.c_void_p, C.c_uint, c_i32p, c_u32p, C.c_void_p]),
            "orc_bp_set_params": (None, [C.c_void_p, c_dp, c_u32p, C.c_double]),
            "orc_bp_get_params": (None, [C.c_void_p, c_dp, c_u32p]),
            "orc_bp_get_state": (None, [C.c_void_p, c_dp, c_dp]),
            "orc_bp_set_state": (None, [C.c_void_p, c_dp, c_dp]),
            "orc_bp_get_h": (None, [C.c_void_p, c_dp]),
            "orc_bp_init_h": (None, [C.c_void_p]),
            "orc_bp_compute_h": (None, [C.c_void_p]),
            "orc_bp_set_field_mix": (None, [C.c_void_p, C.c_double]),
            "orc_bp_set_field_mix_keep": (None, [C.c_void_p, C.c_double]),
            "orc_bp_set_auto_relax": (None, [C.c_void_p, C.c_int]),
            "orc_bp_learn_unconverged": (C.c_int, [C.c_void_p]),
            "orc_bp_set_msg_form": (None, [C.c_void_p, C.c_int]),
            "orc_bp_ar_levels": (None, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "orc_bp_node_update": (C.c_double, [C.c_void_p, C.c_uint32, C.c_double, C.c_int]),
            "orc_bp_converge_async": (C.c_int, [C.c_void_p, C.c_float, C.c_uint, C.c_float, C.c_void_p, C.c_int]),
            "orc_bp_sweep_sync": (C.c_double, [C.c_void_p, C.c_double]),
            "orc_bp_converge_sync": (C.c_int, [C.c_void_p, C.c_double, C.c_uint, C.c_double, c_dp]),
            "orc_bp_free_energy": (C.c_double, [C.c_void_p, C.c_int, c_dp]),
            "orc_bp_entropy": (C.c_double, [C.c_void_p, C.c_int, c_dp]),
            "orc_bp_em_expect": (None, [C.c_void_p, c_dp, c_dp, c_dp]),
            "orc_bp_overlap": (C.c_double, [C.c_void_p]),
            "orc_bp_learning": (C.c_int, [C.c_void_p, C.c_float, C.c_uint, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, c_dp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


class Graph:
    """CSR graph in the engine's layout: row_ptr u64[N+1], nbr u32[E2] ascending, rev u32[E2]."""

    def __init__(self, handle):
        L = lib()
        self._h = handle
        self.N = L.orc_graph_n(handle)
        self.E2 = L.orc_graph_e2(handle)
        self.row_ptr = np.zeros(self.N + 1, dtype=np.uint64)
        self.nbr = np.zeros(self.E2, dtype=np.uint32)
        self.rev = np.zeros(self.E2, dtype=np.uint32)
        L.orc_graph_copy(handle, self.row_ptr.ctypes.data_as(c_u64p), self.nbr.ctypes.data_as(c_u32p),
                         self.rev.ctypes.data_as(c_u32p))

    @classmethod
    def from_edges(cls, pairs, N):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        return cls(lib().orc_graph_from_edges(pairs.ctypes.data_as(c_u32p), pairs.shape[0], N))

    @classmethod
    def from_edgelist(cls, path, N):
        return cls(lib().orc_graph_load_edgelist(os.fsencode(path), N))

    @property
    def deg(self):
        return np.diff(self.row_ptr).astype(np.int64)

    def __del__(self):
        try:
            lib().orc_graph_free(self._h)
        except Exception:
            pass


class Rng:
    def __init__(self, seed):
        self._h = lib().orc_rng_create(seed)

    def draw(self):
        return lib().orc_rng_draw(self._h)

    def __del__(self):
        try:
            lib().orc_rng_free(self._h)
        except Exception:
            pass


def param_from_epsilon_c(N, Q, eps, c):
    cab = np.zeros((Q, Q))
    na = np.zeros(Q, dtype=np.uint32)
    lib().orc_param_from_epsilon_c(N, Q, eps, c, _dp(cab), na.ctypes.data_as(c_u32p))
    return cab, na


def param_from_direct(N, Q, pa, cab_upper):
    pa = np.ascontiguousarray(pa, dtype=np.float64)
    cu = np.ascontiguousarray(cab_upper, dtype=np.float64)
    cab = np.zeros((Q, Q))
    na = np.zeros(Q, dtype=np.uint32)
    lib().orc_param_from_direct(N, Q, _dp(pa), _dp(cu), _dp(cab), na.ctypes.data_as(c_u32p))
    return cab, na


class OracleBP:
    """Mirror of the reference's class belief_propagation on the new layout (see bp_oracle.cpp)."""

    def __init__(self, graph, Q, deg_corr_flag=0):
        self.g = graph
        self.Q = Q
        self.dc = deg_corr_flag
        self._h = lib().orc_bp_create(graph._h, Q, deg_corr_flag)

    def __del__(self):
        try:
            lib().orc_bp_free(self._h)
        except Exception:
            pass

    def init_messages(self, flag, conf, true_conf, rng):
        tc = np.ascontiguousarray(true_conf, dtype=np.uint32)
        cf = None if conf is None else np.ascontiguousarray(conf, dtype=np.int32)
        lib().orc_bp_init_messages(self._h, flag, None if cf is None else cf.ctypes.data_as(c_i32p),
                                   tc.ctypes.data_as(c_u32p), rng._h)

    def set_params(self, cab, na, beta=1.0):
        cab = np.ascontiguousarray(cab, dtype=np.float64)
        na = np.ascontiguousarray(na, dtype=np.uint32)
        lib().orc_bp_set_params(self._h, _dp(cab), na.ctypes.data_as(c_u32p), beta)

    def get_params(self):
        cab = np.zeros((self.Q, self.Q))
        na = np.zeros(self.Q, dtype=np.uint32)
        lib().orc_bp_get_params(self._h, _dp(cab), na.ctypes.data_as(c_u32p))
        return cab, na

    def get_state(self):
        psi = np.zeros((self.g.N, self.Q))
        msg = np.zeros((self.g.E2, self.Q))
        lib().orc_bp_get_state(self._h, _dp(psi), _dp(msg))
        return psi, msg

    def set_state(self, psi=None, msg_out=None):
        psi = None if psi is None else np.ascontiguousarray(psi, dtype=np.float64)
        msg = None if msg_out is None else np.ascontiguousarray(msg_out, dtype=np.float64)
        lib().orc_bp_set_state(self._h, _dp(psi), _dp(msg))

    def h(self):
        h = np.zeros(self.Q)
        lib().orc_bp_get_h(self._h, _dp(h))
        return h

    def init_h(self):
        lib().orc_bp_init_h(self._h)

    def compute_h(self):
        lib().orc_bp_compute_h(self._h)

    def set_field_mix(self, alpha):
        """relaxation of the global field in the synchronous schedule: S <- (1-alpha) S_prev + alpha sum_i g_i psi_i"""
        lib().orc_bp_set_field_mix(self._h, alpha)

    def set_field_mix_keep(self, alpha):
        """as set_field_mix, but the previous sweep's sums stay (a change of the mix in the middle of a run)"""
        lib().orc_bp_set_field_mix_keep(self._h, alpha)

    def set_auto_relax(self, on):
        """adaptive relaxation of converge_sync (the engine's default schedule); off = plain Jacobi sweeps"""
        lib().orc_bp_set_auto_relax(self._h, int(on))

    def set_msg_form(self, on):
        """mirror of sbmbp_set_gather_mode(1): every sweep reports 1-step differences"""
        lib().orc_bp_set_msg_form(self._h, int(on))

    def ar_levels(self):
        """(field level, generic level) the last converge_sync ended on; (0, -1) = it never relaxed"""
        a, b = C.c_int(0), C.c_int(0)
        lib().orc_bp_ar_levels(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def node_update(self, i, damp=1.0, large=False):
        return lib().orc_bp_node_update(self._h, i, damp, int(large))

    def converge_async(self, crit, tmax, damp, rng, conditional=True):
        return lib().orc_bp_converge_async(self._h, crit, tmax, damp, rng._h, int(conditional))

    def sweep_sync(self, damp=1.0):
        return lib().orc_bp_sweep_sync(self._h, damp)

    def converge_sync(self, crit, tmax, damp=1.0):
        last = C.c_double(0.0)
        it = lib().orc_bp_converge_sync(self._h, crit, tmax, damp, C.byref(last))
        return it, last.value

    def free_energy(self, series_K=0):
        parts = np.zeros(3)
        f = lib().orc_bp_free_energy(self._h, series_K, _dp(parts))
        return f, parts

    def entropy(self, series_K=0):
        parts = np.zeros(3)
        e = lib().orc_bp_entropy(self._h, series_K, _dp(parts))
        return e, parts

    def em_expect(self):
        na = np.zeros(self.Q)
        nna = np.zeros(self.Q)
        cab = np.zeros((self.Q, self.Q))
        lib().orc_bp_em_expect(self._h, _dp(na), _dp(nna), _dp(cab))
        return na, nna, cab

    def overlap(self):
        return lib().orc_bp_overlap(self._h)

    def learning(self, lcrit, tmax, lr, damp, rng, sync=False, series_K=0):
        f = C.c_double(0.0)
        steps = lib().orc_bp_learning(self._h, lcrit, tmax, lr, damp, rng._h if rng is not None else None,
                                      int(sync), series_K, C.byref(f))
        return steps, f.value

    def learn_unconverged(self):
        """BP runs inside the last learning() call that hit the sweep limit (0: every run converged)"""
        return lib().orc_bp_learn_unconverged(self._h)
