"""Label counts above 16 (17 .. 64): the matrix-core kernels of csrc/kernels_wide.h against the oracle's synchronous twin.
The reference has no cap on Q (main.cpp:271; loops over Q_ at belief_propagation.cpp:991-1049); up to round 2 the engine
stopped at 16."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import sbm_bp_amd as S
    S.load_library()
    return S


def _instance(Q, dc, seed, N=900, c=6.0, long_row=150):
    rng = np.random.default_rng(4000 + seed)
    tc = (np.arange(N) * Q // N).astype(np.uint32)
    m = int(N * c / 2)
    a = rng.integers(0, N, size=m)
    same = rng.random(m) < 0.85
    b = np.where(same, (a // (N // Q + 1)) * (N // Q + 1) + rng.integers(0, N // Q + 1, size=m), rng.integers(0, N, size=m)) % N
    pairs = np.stack([a, b], 1)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    pairs = pairs[(pairs[:, 0] < N - 5) & (pairs[:, 1] < N - 5)]  # the last five vertices stay isolated
    if long_row:  # one row above the 64-edge segment: the two-walk path of k_wsweep
        pairs = np.concatenate([pairs, np.stack([np.full(long_row, 3), rng.choice(np.arange(10, N - 5), long_row, replace=False)], 1)])
    cab = rng.uniform(0.4, 1.6, size=(Q, Q))
    cab = (cab + cab.T) / 2 + np.eye(Q) * rng.uniform(4.0, 9.0) * Q / 4
    if dc:
        cab = cab / (2.0 * len(pairs) / N) ** 2
    na = np.maximum(1, np.bincount(tc, minlength=Q)).astype(np.uint32)
    return dict(N=N, Q=Q, dc=dc, pairs=pairs.astype(np.uint32), tc=tc, cab=cab, na=na, seed=seed)


def _pair(S, orc, t, flag=0, conf=None, beta=1.0):
    g = S.Graph.from_edges(t["pairs"], t["N"])
    og = orc.Graph.from_edges(t["pairs"], t["N"])
    bp = S.bp_conditional()
    bp.init_messages(S.blockmodel_t(g, t["Q"], t["dc"]), flag, conf, t["tc"], t["seed"])
    bp.set_beta(beta)
    bp.expand_bp_params(S.bp_blockmodel_state(t["cab"], t["na"]))
    ob = orc.OracleBP(og, t["Q"], t["dc"])
    ob.init_messages(flag, conf, t["tc"], orc.Rng(t["seed"]))
    ob.set_params(t["cab"], t["na"], beta)
    ob.set_msg_form(True)  # the wide path reports 1-step differences on every sweep
    return g, bp, ob


@pytest.mark.parametrize("Q,dc", [(17, 0), (32, 0), (32, 1), (48, 0), (64, 0), (64, 1)])
def test_wide_sweeps_and_convergence_against_the_oracle(S, orc, Q, dc):
    t = _instance(Q, dc, seed=Q + dc)
    g, bp, ob = _pair(S, orc, t, beta=0.8 if (Q == 32 and dc == 0) else 1.0)
    assert g.max_degree > 64
    psi0, msg0 = bp.get_state()
    opsi0, omsg0 = ob.get_state()
    assert np.array_equal(psi0, opsi0) and np.array_equal(msg0, omsg0)
    for k in range(4):
        damp = 0.7 if k < 2 else 1.0
        d1, d2 = bp.sweep(1, damp), ob.sweep_sync(damp)
        psi, msg = bp.get_state()
        opsi, omsg = ob.get_state()
        assert abs(d1 - d2) < 1e-11, (k, d1, d2)
        assert np.abs(psi - opsi).max() < 1e-11 and np.abs(msg - omsg).max() < 1e-11, k
    assert abs(bp.compute_overlap() - ob.overlap()) < 1e-11
    # the reductions of -m infer on this (unconverged) state: free energy and entropy, part by part; row sums
    ob.compute_h()
    f, fp = bp.compute_free_energy(parts=True)
    fo, fop = ob.free_energy(0)
    assert np.abs(np.array(fp) - fop).max() <= 1e-9 * max(1.0, np.abs(fop).max()), (fp, fop)
    e, ep = bp.compute_entropy(parts=True)
    eo, eop = ob.entropy(0)
    if dc:
        assert np.isnan(e) and np.isnan(eo)  # the reference prints -nan for deg_corr_flag != 0
    else:
        assert np.abs(np.array(ep) - eop).max() <= 1e-8 * max(1.0, np.abs(eop).max()), (ep, eop)
        bp.set_nonedge_mode(2, 2)  # the moment series (order 2 is all the tensors of Q = 17 .. 64 allow) against the exact pairs
        f2, fp2 = bp.compute_free_energy(parts=True)
        bound = t["N"] * (t["cab"].max() / t["N"]) ** 3 / 6.0  # SURVEY A.4 truncation bound of order 2 (1e-8 and less from N = 1e5 on)
        assert abs(fp2[2] - fp[2]) < 4.0 * bound + 1e-9, (fp2[2], fp[2], bound)
        bp.set_nonedge_mode(0, 0)
    na1, nna1, cab1 = bp.em_expectations()
    na2, nna2, cab2 = ob.em_expect()
    assert np.abs(na1 - na2).max() < 1e-9 and np.abs(nna1 - nna2).max() < 1e-8
    assert np.abs(cab1 - cab2).max() <= 1e-9 * max(1.0, np.abs(cab2).max())  # the numerators: a labels x labels product over the edges (k_wem)
    n1, l1 = bp.converge(1e-10, 600, 1.0)
    n2, l2 = ob.converge_sync(1e-10, 600, 1.0)
    assert n1 == n2, (n1, n2, l1, l2)  # the same sweep - or, on a hard instance, both at the limit with the same last difference
    if n1 >= 0:
        assert l1 < 1e-10
        assert np.abs(bp.get_state()[0] - ob.get_state()[0]).max() < 1e-9
    # (a run that does not converge is a chaotic trajectory: both report -1, their last differences need not agree)
    assert bp.stats().psi_form_sweeps == 0


def test_wide_clamped_rows_and_device_initial_state(S, orc):
    t = _instance(32, 0, seed=7, long_row=100)
    conf = np.where(np.arange(t["N"]) % 9 == 0, t["tc"].astype(np.int32), -1).astype(np.int32)
    conf[3] = int(t["tc"][3])  # the long row is clamped too
    g, bp, ob = _pair(S, orc, t, flag=1, conf=conf)
    psi0 = bp.get_state()[0]
    for _ in range(3):
        assert abs(bp.sweep(1, 1.0) - ob.sweep_sync(1.0)) < 1e-11
    psi, msg = bp.get_state()
    opsi, omsg = ob.get_state()
    assert np.abs(psi - opsi).max() < 1e-11 and np.abs(msg - omsg).max() < 1e-11
    assert np.array_equal(psi[conf != -1], psi0[conf != -1])
    # the device initial state (random marginals, message = sender's marginal) converges to a fixed point of the plain update
    bp2 = S.bp_conditional()
    bp2.init_messages_device(S.blockmodel_t(g, 32, 0), t["tc"], 99)
    bp2.expand_bp_params(S.bp_blockmodel_state(t["cab"], t["na"]))
    n, last = bp2.converge(1e-10, 1000, 1.0)
    assert n >= 0
    p2, m2 = bp2.get_state()
    ob.set_state(p2, m2)
    assert ob.sweep_sync(1.0) < 1e-8


def test_wide_learning_follows_the_oracle(S, orc):
    """-m learn above Q = 16: the EM loop (BP runs, EM expectations through the matrix cores, free energy) step for step
    against the oracle's synchronous EM run"""
    from sbm_bp_amd import synth
    N, Q = 600, 20
    pairs, cin, cout = synth.planted_partition(N, Q, 14.0, 0.02, 33)
    tc = synth.true_conf(N, Q)
    na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)
    cab0 = synth.cab_matrix(Q, 0.85 * cin, 1.5 * cout)
    g = S.Graph.from_edges(pairs, N)
    bm = S.blockmodel_t(g, Q, 0)
    bp = S.bp_basic()
    bp.init_messages(bm, 0, None, tc, 3)
    res = bp.learning(bm, S.bp_blockmodel_state(cab0, na), 1e-6, 40, 0.3, 1.0)
    og = orc.Graph.from_edges(pairs, N)
    ob = orc.OracleBP(og, Q, 0)
    ob.init_messages(0, None, tc, orc.Rng(3))
    ob.set_params(cab0, na, 1.0)
    ob.set_msg_form(True)
    steps, f = ob.learning(1e-6, 40, 0.3, 1.0, None, sync=True, series_K=0)
    cab, na1 = bp.get_params()
    ocab, ona = ob.get_params()
    assert res.em_steps == steps, (res.em_steps, steps)
    assert np.abs(na1.astype(np.int64) - ona.astype(np.int64)).max() <= 1
    if list(na1) == list(ona):
        assert np.abs(cab - ocab).max() < 1e-6 * np.abs(ocab).max() and abs(res.free_energy - f) < 1e-7 * max(1.0, abs(f))


def test_wide_limits_fail_loudly(S):
    t = _instance(20, 0, seed=1, N=200, long_row=0)
    g = S.Graph.from_edges(t["pairs"], t["N"])
    with pytest.raises(Exception):
        S.bp_conditional().init_messages(S.blockmodel_t(g, 65, 0), 0, None, np.zeros(t["N"], dtype=np.uint32), 0)  # above 64
    with pytest.raises(Exception):
        S.bp_conditional().init_messages(S.blockmodel_t(g, 20, 2), 0, None, t["tc"], 0)  # deg_corr_flag 2 above Q = 16
    bp = S.bp_conditional()
    bp.init_messages(S.blockmodel_t(g, 20, 0), 0, None, t["tc"], 0)
    cab = t["cab"].copy()
    cab[0, 1] = cab[1, 0] = 0.0
    with pytest.raises(Exception):
        bp.expand_bp_params(S.bp_blockmodel_state(cab, t["na"]))  # zeros in cab above Q = 16
