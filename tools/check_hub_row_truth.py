"""Verification aid: for fuzz instances with a ~1500-edge row, recompute that row's new marginal in log domain with
numpy (independent of both the engine and the oracle) from the state before each sweep, and say who agrees."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import sbm_bp_amd as S
import oracle as orc
import test_gpu_fuzz as tf


def truth_row(i, row_ptr, nbr, rev, psi, msg, cab, na, dc, beta, N):
    deg = np.diff(row_ptr.astype(np.int64))
    Q = psi.shape[1]
    g = np.ones(N) if dc == 0 else deg.astype(float)
    Ssum = (g[:, None] * psi).sum(0)
    h = cab.T @ Ssum
    lo, hi = int(row_ptr[i]), int(row_ptr[i + 1])
    logp = np.log(na / N) - (beta if dc == 0 else deg[i]) * h / N
    for k in range(lo, hi):
        l = int(nbr[k])
        m_in = msg[int(rev[k])]
        if dc == 0:
            W = cab ** beta
        elif dc == 1:
            W = deg[i] * deg[l] * cab
        else:
            x = deg[i] * deg[l] * cab / N
            W = x / (1 + x)
        b = W.T @ m_in
        with np.errstate(divide="ignore"):
            logp = logp + np.log(b)
    logp -= logp.max()
    p = np.exp(logp)
    return p / p.sum()


for seed in [int(x) for x in sys.argv[1:]] or [60, 96]:
    t = tf._instance(seed)
    Q, N, dc = t["Q"], t["N"], t["dc"]
    g = S.Graph.from_edges(t["pairs"], N)
    og = orc.Graph.from_edges(t["pairs"], N)
    row_ptr, nbr, rev = g.csr()
    hub = int(np.argmax(np.diff(row_ptr.astype(np.int64))))
    bp = S.bp_conditional()
    bp.init_messages(S.blockmodel_t(g, Q, dc), t["flag"], t["conf"], t["tc"], t["seed"])
    bp.set_beta(t["beta"])
    bp.expand_bp_params(S.bp_blockmodel_state(t["cab"], t["na"]))
    ob = orc.OracleBP(og, Q, dc)
    ob.init_messages(t["flag"], t["conf"], t["tc"], orc.Rng(t["seed"]))
    ob.set_params(t["cab"], t["na"], t["beta"])
    print("seed %d: Q=%d dc=%d beta=%g hub row %d degree %d clamped=%s" % (seed, Q, dc, t["beta"], hub, row_ptr[hub + 1] - row_ptr[hub],
          None if t["conf"] is None else t["conf"][hub]))
    for k in range(4):
        damp = t["damp"] if k < 2 else 1.0
        psi_e, msg_e = bp.get_state()
        psi_o, msg_o = ob.get_state()
        want_e = truth_row(hub, row_ptr, nbr, rev, psi_e, msg_e, t["cab"], t["na"].astype(float), dc, t["beta"], N)
        want_o = truth_row(hub, row_ptr, nbr, rev, psi_o, msg_o, t["cab"], t["na"].astype(float), dc, t["beta"], N) if np.isfinite(msg_o).all() else None
        bp.sweep(1, damp)
        ob.sweep_sync(damp)
        got_e, got_o = bp.get_state()[0][hub], ob.get_state()[0][hub]
        print("  sweep %d: engine vs truth(from engine state) %.2e | oracle vs truth(from oracle state) %s | engine-oracle state diff before sweep %.2e" % (
            k, np.abs(got_e - want_e).max(), "n/a" if want_o is None else "%.2e" % np.abs(got_o - want_o).max(),
            np.nanmax(np.abs(msg_e - msg_o)) if msg_e.size else 0.0), flush=True)
