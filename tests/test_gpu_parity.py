"""Parity of the HIP engine (through the C ABI) against the oracle and the committed reference
goldens. Floating point: the engine must land on the reference's fixed point; tolerances are the
north star's 1e-5 relative, tightened to what fp64 actually delivers (stated per assert)."""
import os

import numpy as np
import pytest

from conftest import args_of, best_perm_diff, best_vector_perm, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import sbm_bp_amd as S
    S.load_library()
    return S


def engine_from(S, a, learn=False, seed=None):
    g = S.load_edge_list(a["path"], a["N"])
    bm = S.blockmodel_t(g, a["Q"], a["dc"])
    bp = S.bp_basic() if learn else S.bp_conditional()
    bp.init_messages(bm, a["init_flag"], a.get("beliefs"), a["true_conf"], a["seed"] if seed is None else seed)
    bp.set_beta(a["beta"])
    if "eps" in a:
        st = S.bp_param_from_epsilon_c(bm, a["eps"], a["c"])
    else:
        st = S.bp_param_from_direct(bm, a["pa"], a["cab_upper"])
    bp.expand_bp_params(st)
    return g, bm, bp, st


def oracle_from(orc, a):
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    bp = orc.OracleBP(g, a["Q"], a["dc"])
    rng = orc.Rng(a["seed"])
    bp.init_messages(a["init_flag"], a.get("beliefs"), a["true_conf"], rng)
    if "eps" in a:
        cab, na = orc.param_from_epsilon_c(a["N"], a["Q"], a["eps"], a["c"])
    else:
        cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    bp.set_params(cab, na, a["beta"])
    return g, bp, rng


TIGHT = ["c1_matched_tight_seed0", "c1_matched_beta08_seed0", "c1_dc1_tight_seed0", "c1_dc2_tight_seed0",
         "q4_tight_seed0", "c1_planted_i1_seed0", "hub_dc1_tight_seed0", "q10_tight_seed1"]


@pytest.mark.parametrize("name", TIGHT + ["hub_dc0_tight_seed0"])
def test_initial_state_is_the_reference_rng_stream(S, orc, name):
    a = args_of(golden(name))
    _, _, bp, _ = engine_from(S, a)
    _, obp, _ = oracle_from(orc, a)
    psi, msg = bp.get_state()
    opsi, omsg = obp.get_state()
    # bit-exact: same mt19937 draws, same fill order. The device keeps Q-1 components of every message and restores the
    # largest as 1 - sum of the others, so that ONE entry per message comes back within a few ulp instead of bit-exact
    assert (psi == opsi).all()
    assert ((msg != omsg).sum(1) <= 1).all() and np.abs(msg - omsg).max() < 5e-16
    rows = np.flatnonzero((msg != omsg).any(1))
    assert (np.argmax(omsg[rows], 1) == np.argmax(msg[rows] != omsg[rows], 1)).all()  # only the largest component moves


@pytest.mark.parametrize("name", TIGHT + ["hub_dc0_tight_seed0"])
def test_sweeps_match_the_synchronous_oracle(S, orc, name):
    a = args_of(golden(name))
    _, _, bp, _ = engine_from(S, a)
    _, obp, _ = oracle_from(orc, a)
    for sweep in range(3):
        d_gpu = bp.sweep(1, 1.0)
        d_cpu = obp.sweep_sync(1.0)
        psi, msg = bp.get_state()
        opsi, omsg = obp.get_state()
        assert np.abs(msg - omsg).max() < 1e-12, "messages after sweep %d" % sweep
        assert np.abs(psi - opsi).max() < 1e-12
        assert abs(d_gpu - d_cpu) < 1e-12
    assert np.abs(msg.sum(1) - 1).max() < 1e-14 and np.abs(psi.sum(1) - 1).max() < 1e-14


def test_damped_sweeps_match_the_oracle(S, orc):
    a = args_of(golden("c1_matched_damped_seed0"))
    _, _, bp, _ = engine_from(S, a)
    _, obp, _ = oracle_from(orc, a)
    for _ in range(4):
        d_gpu, d_cpu = bp.sweep(1, 0.5), obp.sweep_sync(0.5)
        assert abs(d_gpu - d_cpu) < 1e-12
    assert np.abs(bp.get_state()[1] - obp.get_state()[1]).max() < 1e-12


@pytest.mark.parametrize("name", TIGHT)
def test_converged_fixed_point_equals_reference_golden(S, name):
    gd = golden(name)
    a, r = args_of(gd), gd["result"]
    _, _, bp, _ = engine_from(S, a)
    niter, last = bp.converge(1e-13, 5000, 1.0)
    assert niter >= 0 and last < 1e-13
    psi = bp.real_psi()
    d, perm = best_perm_diff(psi, np.array(r["psi"]).reshape(psi.shape))
    assert d < 1e-9  # marginals modulo label permutation; north star asks 1e-5 relative
    f, parts = bp.compute_free_energy(parts=True)
    assert abs(f - r["f"]) <= 1e-9 * max(1.0, abs(r["f"]))  # north star: 1e-5 relative
    for got, key in zip(parts, ("f_site", "f_edge", "f_nonedge")):
        assert abs(got - r[key]) <= 1e-9 * max(1.0, abs(r[key]))
    if a["Q"] <= 8:
        assert abs(bp.compute_overlap() - r["overlap"]) < 1e-9
    else:  # above Q = 8 the reference scores the identity labelling only (bp.cpp:784-790): relabel, then compare
        tc = a["true_conf"]
        assert abs(psi[:, list(perm)][np.arange(a["N"]), tc].sum() / a["N"] - r["overlap"]) < 1e-9
        C = bp.confusion()
        assert abs(bp.compute_overlap() - np.trace(C) / a["N"]) < 1e-12
    e, eparts = bp.compute_entropy(parts=True)
    if np.isnan(r["e"]):
        assert np.isnan(e)  # reference prints -nan for deg_corr_flag != 0
    else:
        assert abs(e - r["e"]) <= 1e-9 * max(1.0, abs(r["e"]))
        for got, key in zip(eparts, ("e_site", "e_edge", "e_nonedge")):
            assert abs(got - r[key]) <= 1e-9 * max(1.0, abs(r[key]))


def test_default_schedule_converges_on_the_hub_graph_where_the_reference_does(S, orc):
    """plain SBM on a power-law graph: pure Jacobi oscillates through the lagged global field (the reference converges in 139
    sweeps). With NOTHING but the reference's flags the engine sees the swing of the field sums, lowers the field mix on the
    device and reaches the reference's fixed point, on the same sweep as the oracle's synchronous twin."""
    gd = golden("hub_dc0_tight_seed0")
    a, r = args_of(gd), gd["result"]
    _, _, bp, _ = engine_from(S, a)
    niter, last = bp.converge(a["crit"], a["tmax"], 1.0)  # the fixture's own -e 1e-12 -t 2000
    assert niter >= 0 and last < a["crit"]
    fl, gl, mix, dmp = bp.relaxation()
    assert (fl, gl, mix, dmp) == (1, -1, 0.25, 1.0)
    _, obp, _ = oracle_from(orc, a)
    it_cpu, _ = obp.converge_sync(a["crit"], a["tmax"], 1.0)
    assert obp.ar_levels() == (1, -1) and abs(niter - it_cpu) <= 2 and abs(niter - r["niter"]) <= 10  # reference: 139
    psi = bp.real_psi()
    d, _ = best_perm_diff(psi, np.array(r["psi"]).reshape(psi.shape))
    assert d < 1e-9
    f, parts = bp.compute_free_energy(parts=True)
    assert abs(f - r["f"]) <= 1e-9 * abs(r["f"])
    assert abs(bp.compute_overlap() - r["overlap"]) < 1e-9
    # plain Jacobi (adaptive relaxation off) does not converge here; a fixed mix of 0.1 reaches the same fixed point
    _, _, bp, _ = engine_from(S, a)
    bp.set_auto_relax(False)
    niter, last = bp.converge(1e-13, 300, 1.0)
    assert niter == -1 and last > 0.1 and bp.relaxation()[:2] == (0, -1)
    _, _, bp, _ = engine_from(S, a)
    bp.set_schedule(field_mix=0.1, check_every=8)
    niter, last = bp.converge(1e-13, 5000, 1.0)
    assert niter >= 0
    psi = bp.real_psi()
    assert best_perm_diff(psi, np.array(r["psi"]).reshape(psi.shape))[0] < 1e-9


def test_relaxed_run_ends_on_a_fixed_point_the_reference_reaches(S, orc):
    """the hub instance has three BP fixed points and the reference's own answer depends on its seed (fixtures
    hub_dc0_tight_seed0 / _seed1 / _seed23: 26, 13 and 2 of the reference's seeds 0..40). Whatever seed the engine starts
    from, where it converges is one of the three, to 1e-9 in the free energy and 1e-8 in the marginals"""
    gs = [golden("hub_dc0_tight_seed%d" % d) for d in (0, 1, 23)]
    a = args_of(gs[0])
    seen = set()
    for seed in range(10):
        _, _, bp, _ = engine_from(S, a, seed=seed)
        niter, last = bp.converge(a["crit"], a["tmax"], 1.0)
        assert niter >= 0, seed
        psi = bp.real_psi()
        f = bp.compute_free_energy()
        for k, gd in enumerate(gs):
            if abs(f - gd["result"]["f"]) <= 1e-9 * abs(f):
                assert best_perm_diff(psi, np.array(gd["result"]["psi"]).reshape(psi.shape))[0] < 1e-8
                seen.add(k)
                break
        else:
            raise AssertionError("seed %d converged to f = %.12f, which neither reference run reaches" % (seed, f))
    assert len(seen) >= 2  # (which seed goes where is schedule dependent, in the reference too)


@pytest.mark.parametrize("name", ["c1_matched_tight_seed0", "q4_tight_seed0", "c1_dc1_tight_seed0", "hub_dc1_tight_seed0", "q10_tight_seed1",
                                  "c1_planted_i1_seed0"])
def test_fused_reduction_pass_equals_the_separate_kernels(S, name):
    """after a converge call the free energy, the entropy and the EM expectations come out of ONE pass on the marginal-gather
    reconstruction (k_fe_psi); with the message-gather mode forced they come from the round-2 kernels (k_fe_frame,
    k_nonedge_*_adj, k_em_edges), which gather the incoming messages through rev. Same state, same numbers."""
    a = args_of(golden(name))
    _, _, bp, _ = engine_from(S, a)
    niter, _ = bp.converge(1e-10, 3000, 1.0)
    assert niter >= 0
    def everything():
        f, fp = bp.compute_free_energy(parts=True)
        e, ep = bp.compute_entropy(parts=True)
        return np.array([f] + list(fp) + [e] + list(ep)), np.concatenate([np.ravel(x) for x in bp.em_expectations()])
    fused, em_fused = everything()
    bp.set_gather_mode(1)
    sep, em_sep = everything()
    ok = np.isfinite(sep)
    assert (np.isfinite(fused) == ok).all()
    assert np.abs(fused[ok] - sep[ok]).max() <= 1e-11 * max(1.0, np.abs(sep[ok]).max()), (fused, sep)
    assert np.abs(em_fused - em_sep).max() <= 1e-10 * max(1.0, np.abs(em_sep).max())


@pytest.mark.parametrize("flag", [2, 3])
def test_init_flags_2_and_3_run_against_the_oracle(S, orc, flag):
    """-i 2 (planted labels plus noise) and -i 3 (hard planted labels): the reference aborts on an assert for any vertex planted
    in group 1, so there is no fixture; engine and oracle share the sane definition (oracle/bp_oracle.cpp
    init_messages). Here they RUN: identical initial state, three sweeps to 1e-12, then converge to the same sweep and the same
    free energy - with the clamped rows of the beliefs file (bp_conditional) constant throughout."""
    a = dict(args_of(golden("c1_planted_i1_seed0")), init_flag=flag)
    _, _, bp, _ = engine_from(S, a)
    _, obp, _ = oracle_from(orc, a)
    psi0, msg0 = bp.get_state()
    opsi0, omsg0 = obp.get_state()
    # (messages live as records of Q-1 components on the device: the largest one comes back as 1 - sum, an ulp off at most)
    assert np.array_equal(psi0, opsi0) and np.abs(msg0 - omsg0).max() < 4e-16
    clamped = np.flatnonzero(a["beliefs"] != -1)
    for _ in range(3):
        assert abs(bp.sweep(1, 1.0) - obp.sweep_sync(1.0)) < 1e-12
        psi, msg = bp.get_state()
        opsi, omsg = obp.get_state()
        assert np.abs(psi - opsi).max() < 1e-12 and np.abs(msg - omsg).max() < 1e-12
        assert np.array_equal(psi[clamped], psi0[clamped])  # clamped rows never move (bp.cpp:1115-1124)
    n1, l1 = bp.converge(1e-10, 2000, 1.0)
    n2, l2 = obp.converge_sync(1e-10, 2000, 1.0)
    assert n1 >= 0 and n2 - 1 <= n1 <= n2 + 8
    assert abs(bp.compute_free_energy() - obp.free_energy(0)[0]) < 1e-9
    assert abs(bp.compute_overlap() - obp.overlap()) < 1e-9


def test_niter_and_batched_convergence_check_agree_with_oracle(S, orc):
    a = args_of(golden("c1_matched_default_seed0"))
    _, obp, _ = oracle_from(orc, a)
    it_cpu, _ = obp.converge_sync(a["crit"], a["tmax"], 1.0)
    for every in (1, 7):
        _, _, bp, _ = engine_from(S, a)
        bp.set_schedule(1.0, every)
        it_gpu, _ = bp.converge(a["crit"], a["tmax"], 1.0)
        assert it_gpu == it_cpu  # the device-side stop flag makes niter independent of the batch size
        assert np.abs(bp.get_state()[1] - obp.get_state()[1]).max() < 1e-11
    # the reference's asynchronous niter for this run is 56; sync needs a comparable number
    assert abs(it_gpu - golden("c1_matched_default_seed0")["result"]["niter"]) <= 10


def test_readme_infer_line(S):
    """README.md:36 command: stdout `2.99556 -0.143476 0.5 <niter>` (niter is schedule dependent)"""
    gd = golden("c1_readme_infer_seed0")
    a, r = args_of(gd), gd["result"]
    g, bm, bp, st = engine_from(S, a)
    res = bp.inference(bm, st, a["crit"], a["tmax"], a["damp"])
    from sbm_bp_amd.bp import format_infer_line
    line = format_infer_line(res).split()
    assert line[:2] == ["2.99556", "-0.143476"] and len(line) == 4
    # at the loose default criterion 5e-6 both codes sit within ~criterion of the fixed point
    assert abs(res.overlap - r["overlap"]) < 1e-5
    assert abs(res.free_energy - r["f"]) < 1e-6 and abs(res.entropy - r["e"]) < 1e-6


@pytest.mark.parametrize("name", ["c1_em_expect_seed0", "c1_em_expect_dc1_seed0", "c1_em_expect_dc2_seed0", "q4_em_expect_seed0",
                                  "q10_em_expect_seed1"])
def test_em_expectations_on_fixed_point(S, name):
    gd = golden(name)
    a, r = args_of(gd), gd["result"]
    _, _, bp, _ = engine_from(S, a, learn=True)
    niter, _ = bp.converge(1e-13, 5000, 1.0)
    assert niter >= 0
    na, nna, cab = bp.em_expectations()
    Q = a["Q"]
    ref_na, ref_nna, ref_cab = np.array(r["na_expect"]), np.array(r["nna_expect"]), np.array(r["cab_expect"]).reshape(Q, Q)
    p = best_vector_perm(na, ref_na)
    assert np.abs(na[p] - ref_na).max() < 1e-7 and np.abs(nna[p] - ref_nna).max() < 1e-6
    assert np.abs(cab[np.ix_(p, p)] - ref_cab).max() < 1e-8 * max(1.0, np.abs(ref_cab).max())


@pytest.mark.parametrize("name", ["c1_learn_515_seed0", "q4_learn_seed2", "c1_readme_learn_seed0"])
def test_learning_matches_synchronous_oracle(S, orc, name):
    """EM trajectories depend on the schedule; the engine must follow the oracle's synchronous EM run
    step for step and end near the reference's (asynchronous) learned parameters."""
    gd = golden(name)
    a, r = args_of(gd), gd["result"]
    g, bm, bp, st = engine_from(S, a, learn=True)
    res = bp.learning(bm, st, a["lcrit"], a["tmax"], a["lr"], a["damp"])
    _, obp, _ = oracle_from(orc, a)
    steps, f = obp.learning(a["lcrit"], a["tmax"], a["lr"], a["damp"], None, sync=True, series_K=0)
    cab, na = bp.get_params()
    ocab, ona = obp.get_params()
    assert res.em_steps == steps and list(na) == list(ona)
    # EM amplifies rounding differences over tens of steps: 1e-6 relative, far inside the north star's 1e-5
    assert np.abs(cab - ocab).max() < 1e-6 * np.abs(ocab).max() and abs(res.free_energy - f) < 1e-8
    # ... and against the REFERENCE's own run (asynchronous schedule) of the same command:
    ref_cab = np.array(r["cab_final"]).reshape(cab.shape)
    if name in ("c1_learn_515_seed0", "c1_readme_learn_seed0"):
        # well-posed instances: same integers, parameters to 1e-7 relative (measured 5e-8 / 5e-9)
        assert list(na) == list(r["na_final"])
        assert np.abs(cab - ref_cab).max() < 1e-7 * np.abs(ref_cab).max()
        assert abs(res.overlap - r["overlap"]) < 1e-7
    else:
        # q4_learn_seed2 starts far from the planted parameters and passes a symmetric saddle of BP (two planted groups in
        # one label); plain Jacobi sweeps stay there (f = -2.517), the relaxed field of the EM loop leaves it as the
        # reference does. Same optimum; the integer truncation of na differs along the way.
        _, oref, rng = oracle_from(orc, a)
        oref.learning(a["lcrit"], a["tmax"], a["lr"], a["damp"], rng, sync=False, series_K=0)  # the reference, bit for bit
        f_ref, _ = oref.free_energy(0)
        assert abs(f_ref - (-2.7162086)) < 1e-6
        assert abs(res.free_energy - f_ref) < 1e-3 and abs(res.overlap - r["overlap"]) < 5e-3
        assert np.abs(np.array(na, dtype=np.int64) - np.array(r["na_final"])).max() <= 4
        assert np.abs(np.diag(cab) - np.diag(ref_cab)).max() < 0.15 * np.abs(ref_cab).max()
        # The END POINT of an EM run is schedule dependent here (1e-5 on it is not met on this fixture: |df| = 2.4e-4, the group
        # sizes differ by up to 4 vertices because every EM step truncates them along a different path - DESIGN.md section 2).
        # What IS schedule independent: at the reference's own final parameters, from the reference's own final state, BP has
        # one fixed point, and the engine's free energy, overlap and EM expectations there are the reference's to 1e-9.
        cab_f, na_f = oref.get_params()
        assert list(cab_f.ravel()) == r["cab_final"] and list(na_f) == r["na_final"]  # the replay IS the reference's run
        psi_f, msg_f = oref.get_state()
        assert oref.converge_async(1e-13, 5000, 1.0, rng, False) >= 0  # tightened at those parameters, the reference's schedule
        f_star, _ = oref.free_energy(0)
        bp2 = S.bp_basic()
        bp2.init_messages(bm, a["init_flag"], a.get("beliefs"), a["true_conf"], a["seed"])
        bp2.expand_bp_params(S.bp_blockmodel_state(cab_f, na_f))
        bp2.set_state(psi_f, msg_f)
        niter2, last2 = bp2.converge(1e-13, 5000, 1.0)
        assert niter2 >= 0
        assert abs(bp2.compute_free_energy() - f_star) <= 1e-9 * abs(f_star)
        assert abs(bp2.compute_overlap() - oref.overlap()) < 1e-9
        for got, want in zip(bp2.em_expectations(), oref.em_expect()):
            assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
        assert abs(f_star - f_ref) < 1e-6 and abs(oref.overlap() - r["overlap"]) < 1e-6  # (the reference stopped at its EM criterion)


def test_series_and_exact_nonedge_agree(S):
    a = args_of(golden("c1_matched_tight_seed0"))
    _, _, bp, _ = engine_from(S, a)
    bp.converge(1e-12, 2000, 1.0)
    bp.set_nonedge_mode(1, 0)
    f_exact, p_exact = bp.compute_free_energy(parts=True)
    e_exact, q_exact = bp.compute_entropy(parts=True)
    bp.set_nonedge_mode(2, 4)
    f_ser, p_ser = bp.compute_free_energy(parts=True)
    e_ser, q_ser = bp.compute_entropy(parts=True)
    assert abs(p_exact[2] - p_ser[2]) < 1e-9  # K=4 at N=1000 leaves 1.3e-10
    assert abs(q_exact[2] - q_ser[2]) < 1e-8


def _synthetic(S, orc, N, Q, c, eps, seed, dc=0):
    from sbm_bp_amd import synth
    pairs, cin, cout = synth.planted_partition(N, Q, c, eps, seed)
    g = S.Graph.from_edges(pairs, N)
    og = orc.Graph.from_edges(pairs, N)
    return g, og, synth.cab_matrix(Q, cin, cout), synth.true_conf(N, Q)


@pytest.mark.parametrize("N,Q,c", [(20000, 2, 3.0), (20000, 4, 10.0), (6000, 8, 12.0), (5000, 3, 6.0), (5000, 5, 8.0)])
def test_synthetic_graphs_sweeps_and_reductions(S, orc, N, Q, c):
    g, og, cab, tc = _synthetic(S, orc, N, Q, c, 0.1, 3)
    rp, nbr, rev = g.csr()
    assert (rp == og.row_ptr).all() and (nbr == og.nbr).all() and (rev == og.rev).all()
    na = np.array([N // Q] * Q, dtype=np.uint32)
    bm = S.blockmodel_t(g, Q, 0)
    bp = S.bp_conditional()
    bp.init_messages(bm, 0, None, tc, 5)
    bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
    obp = orc.OracleBP(og, Q, 0)
    obp.init_messages(0, None, tc, orc.Rng(5))
    obp.set_params(cab, na, 1.0)
    for _ in range(5):
        assert abs(bp.sweep(1, 1.0) - obp.sweep_sync(1.0)) < 1e-12
    psi, msg = bp.get_state()
    opsi, omsg = obp.get_state()
    assert np.abs(msg - omsg).max() < 1e-12 and np.abs(psi - opsi).max() < 1e-12
    obp.compute_h()
    f, parts = bp.compute_free_energy(parts=True)
    of, oparts = obp.free_energy(0 if N <= 20000 else 3)
    assert np.abs(parts - oparts).max() < 1e-10 * max(1.0, np.abs(oparts).max())
    na_e, nna_e, cab_e = bp.em_expectations()
    ona, onna, ocab = obp.em_expect()
    assert np.abs(na_e - ona).max() < 1e-8 and np.abs(cab_e - ocab).max() < 1e-9 * max(1.0, np.abs(ocab).max())
    assert abs(bp.compute_overlap() - obp.overlap()) < 1e-12


def test_full_size_properties_c2(S):
    """BASELINE config C2 (N=1e6, Q=2, c=3) through size-independent properties: normalisation,
    label-permutation equivariance, monotone approach to the fixed point, reproducibility."""
    from sbm_bp_amd import synth
    N, Q = 1000000, 2
    pairs, cin, cout = synth.planted_partition(N, Q, 3.0, 0.1, 1)
    g = S.Graph.from_edges(pairs, N)
    tc = synth.true_conf(N, Q)
    cab = synth.cab_matrix(Q, cin, cout)
    na = np.array([N // Q] * Q, dtype=np.uint32)
    bm = S.blockmodel_t(g, Q, 0)
    outs = []
    for rep in range(2):
        bp = S.bp_conditional()
        bp.init_messages_device(bm, tc, 42)
        bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
        diffs = [bp.sweep(1, 1.0) for _ in range(12)]
        psi, msg = bp.get_state()
        outs.append((psi, msg, diffs))
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()  # bitwise reproducible
    psi, msg, diffs = outs[0]
    assert np.abs(msg.sum(1) - 1).max() < 1e-14 and np.abs(psi.sum(1) - 1).max() < 1e-14
    assert msg.min() >= 0 and np.isfinite(msg).all()
    # label permutation: swapping the two labels of the initial state and of cab swaps the result
    bp = S.bp_conditional()
    bp.init_messages_device(bm, tc, 42)
    p0, m0 = bp.get_state()
    bp.set_state(p0[:, ::-1].copy(), m0[:, ::-1].copy())
    bp.expand_bp_params(S.bp_blockmodel_state(cab[::-1, ::-1].copy(), na[::-1].copy()))
    for _ in range(12):
        bp.sweep(1, 1.0)
    p1, m1 = bp.get_state()
    assert np.abs(p1[:, ::-1] - psi).max() < 1e-12 and np.abs(m1[:, ::-1] - msg).max() < 1e-12
    st = bp.stats()
    assert st.edge_msg_updates == 12 * g.E2
    # the run converges at the default criterion and detects the planted groups (C1-like parameters: ~0.84)
    bp.set_schedule(1.0, 8)
    niter, last = bp.converge(5e-6, 1000, 1.0)
    assert niter >= 0 and last < 5e-6
    assert 0.80 < bp.compute_overlap() < 0.88
    f = bp.compute_free_energy()
    assert np.isfinite(f)


def test_full_size_properties_c3(S):
    """BASELINE's headline configuration C3 (N=1e7, Q=4, c=10: 1e8 directed edges) through size-independent properties:
    bitwise reproducibility, the two sweep forms agreeing, field consistency, convergence in the documented number of
    sweeps with the planted groups recovered. (State is compared through reductions: 3.5 GB of messages stay on the device.)"""
    from sbm_bp_amd import synth
    N, Q = 10_000_000, 4
    pairs, cin, cout = synth.planted_partition(N, Q, 10.0, 0.1, 2)
    g = S.Graph.from_edges(pairs, N)
    del pairs
    assert g.E2 > 99_000_000
    tc = synth.true_conf(N, Q)
    st = S.bp_blockmodel_state(synth.cab_matrix(Q, cin, cout), np.array([N // Q] * Q, dtype=np.uint32))
    bm = S.blockmodel_t(g, Q, 0)
    runs = []
    for mode in (0, 0, 1):  # marginal gather twice, then message gather
        bp = S.bp_conditional()
        bp.init_messages_device(bm, tc, 1234)
        bp.expand_bp_params(st)
        bp.set_gather_mode(mode)
        diffs = [bp.sweep(1, 1.0) for _ in range(4)]
        runs.append((diffs, bp.compute_overlap(), bp.compute_free_energy(), bp.confusion(), bp.h().copy()))
        assert bp.stats().psi_form_sweeps == (4 if mode == 0 else 0)  # the device initial state (message = sender's marginal) needs no explicit first sweep
        if mode == 1:
            last = bp
        else:
            del bp
    (d0, o0, f0, c0, h0), (d1, o1, f1, c1, h1), (d2, o2, f2, c2, h2) = runs
    assert d0 == d1 and o0 == o1 and f0 == f1 and (c0 == c1).all() and (h0 == h1).all()  # run to run: bit for bit
    assert np.abs(np.array(d0) - np.array(d2)).max() < 1e-12 and abs(o0 - o2) < 1e-12 and abs(f0 - f2) < 1e-10 * abs(f0)
    assert np.abs(c0 - c2).max() < 1e-6 * N and abs(c0.sum() - N) < 1e-6 * N  # the marginals sum to one per vertex
    niter, diff = last.converge(5e-6, 200, 1.0)
    assert 25 <= niter + 4 + 1 <= 45 and diff < 5e-6  # 35 sweeps from this start
    assert last.compute_overlap() > 0.97


def test_full_size_properties_c4_c5(S):
    """the other two BASELINE configurations at full size. C4 (power-law DC-SBM, Q=8, N=1e6, rows of up to ~2500 edges: segment,
    wave-product and hub kernels in one sweep): reproducible, both sweep forms agree. C5 (N=1e6, Q=4, c=5): -m learn from
    (0.9 cin, 1.8 cout) recovers the planted parameters."""
    from sbm_bp_amd import synth
    N, Q = 1_000_000, 8
    pairs, cab, _ = synth.dc_sbm_powerlaw(N, Q, 8.0, 0.1, 3)
    g = S.Graph.from_edges(pairs, N)
    del pairs
    tc = synth.true_conf(N, Q)
    bm = S.blockmodel_t(g, Q, 1)
    st = S.bp_blockmodel_state(cab, np.array([N // Q] * Q, dtype=np.uint32))
    runs = []
    for mode in (0, 0, 1):
        bp = S.bp_conditional()
        bp.init_messages_device(bm, tc, 99)
        bp.expand_bp_params(st)
        bp.set_gather_mode(mode)
        assert bp.stats().n_hub_rows > 100
        d = [bp.sweep(1, 1.0) for _ in range(4)]
        runs.append((d, bp.compute_overlap(), bp.compute_free_energy(), bp.get_state()[0]))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2] and (runs[0][3] == runs[1][3]).all()
    assert np.abs(np.array(runs[0][0]) - np.array(runs[2][0])).max() < 1e-11 and np.abs(runs[0][3] - runs[2][3]).max() < 1e-11
    assert abs(runs[0][2] - runs[2][2]) < 1e-10 * abs(runs[0][2]) and np.abs(runs[0][3].sum(1) - 1).max() < 1e-14
    del runs, bp, g, bm
    N, Q = 1_000_000, 4
    pairs, cin, cout = synth.planted_partition(N, Q, 5.0, 0.1, 4)
    g = S.Graph.from_edges(pairs, N)
    del pairs
    bm = S.blockmodel_t(g, Q, 0)
    bp = S.bp_basic()
    bp.init_messages_device(bm, synth.true_conf(N, Q), 7)
    start = S.bp_blockmodel_state(synth.cab_matrix(Q, 0.9 * cin, 1.8 * cout), np.array([N // Q] * Q, dtype=np.uint32))
    res = bp.learning(bm, start, 1e-6, 100, 0.2, 1.0)
    learned, _ = bp.get_params()
    assert res.status == 1 and res.em_steps < 60 and res.overlap > 0.85
    assert abs(np.diag(learned).mean() - cin) < 0.01 * cin and abs((learned.sum() - np.trace(learned)) / (Q * Q - Q) - cout) < 0.02 * cout


def test_marginal_gather_and_message_gather_forms_agree(S):
    """the two forms of the sweep kernel produce the same iterates, niter and fixed point"""
    a = args_of(golden("q4_tight_seed0"))
    out = []
    for mode in (0, 1):
        _, _, bp, _ = engine_from(S, a)
        bp.set_gather_mode(mode)
        d5 = bp.sweep(5, 1.0)
        psi5, msg5 = bp.get_state()
        niter, last = bp.converge(1e-12, 3000, 1.0)
        out.append((d5, psi5, msg5, niter, last, bp.real_psi(), bp.compute_free_energy(), bp.stats().psi_form_sweeps))
    a0, a1 = out
    assert a0[7] > 0 and a1[7] == 0  # mode 0 really ran k_sweep_psi, mode 1 never did
    assert abs(a0[0] - a1[0]) < 1e-12 and np.abs(a0[1] - a1[1]).max() < 1e-12 and np.abs(a0[2] - a1[2]).max() < 1e-12
    assert a0[3] == a1[3] and a0[4] < 1e-12 and a1[4] < 1e-12
    assert np.abs(a0[5] - a1[5]).max() < 1e-11 and abs(a0[6] - a1[6]) < 1e-11


def test_extreme_hub_row_in_fragments(S, orc):
    """one vertex joined to 20 000 others (79 fragments of 256 edges for the marginal-gather sweep) on top of a sparse planted
    partition: sweeps, marginals, messages and the free energy against the oracle, with and without degree correction, both
    sweep forms, and with the hub clamped (its fragments copy the marginal and leave the messages alone)"""
    from sbm_bp_amd import synth
    N, Q = 30000, 3
    pairs, cin, cout = synth.planted_partition(N, Q, 4.0, 0.2, 11)
    rng = np.random.default_rng(5)
    hub = 12345
    others = rng.choice(np.delete(np.arange(N), hub), size=20000, replace=False).astype(np.uint32)
    star = np.stack([np.minimum(others, hub), np.maximum(others, hub)], axis=1).astype(np.uint32)
    key = np.unique(np.concatenate([pairs[:, 0].astype(np.int64) * N + pairs[:, 1], star[:, 0].astype(np.int64) * N + star[:, 1]]))
    pairs = np.stack([key // N, key % N], axis=1).astype(np.uint32)
    g = S.Graph.from_edges(pairs, N)
    og = orc.Graph.from_edges(pairs, N)
    assert g.max_degree >= 20000
    tc = synth.true_conf(N, Q)
    na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)
    c_eff = 2.0 * len(pairs) / N
    for dc, cab in ((0, synth.cab_matrix(Q, cin, cout)), (1, synth.cab_matrix(Q, cin, cout) / (c_eff * c_eff))):
        obp = orc.OracleBP(og, Q, dc)
        obp.init_messages(0, None, tc, orc.Rng(3))
        obp.set_params(cab, na, 1.0)
        od = [obp.sweep_sync(1.0) for _ in range(5)]
        opsi, omsg = obp.get_state()
        obp.compute_h()  # the field of the CURRENT marginals, as the engine (and the reference's incremental h) evaluates f_site with
        ofe, _ = obp.free_energy(0)
        for mode in (0, 1):
            bp = S.bp_conditional()
            bp.init_messages(S.blockmodel_t(g, Q, dc), 0, None, tc, 3)
            bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
            bp.set_gather_mode(mode)
            assert bp.stats().n_hub_rows == 1 and bp.stats().hub_edges >= 20000
            d = [bp.sweep(1, 1.0) for _ in range(5)]
            psi, msg = bp.get_state()
            assert np.abs(np.array(d) - np.array(od)).max() < 1e-9, (dc, mode, d, od)
            assert np.abs(psi - opsi).max() < 1e-9 and np.abs(msg - omsg).max() < 1e-9, (dc, mode)
            assert abs(bp.compute_free_energy() - ofe) < 1e-9 * max(1.0, abs(ofe)), (dc, mode)
            assert (bp.stats().psi_form_sweeps > 0) == (mode == 0)
    # the hub clamped (-i 1 style beliefs): marginal-gather form only runs without clamps on the hub's NEIGHBOURS being special,
    # so compare the engine's own two forms and the oracle
    beliefs = np.full(N, -1, dtype=np.int32)
    beliefs[hub] = 1
    beliefs[7] = 0
    cab = synth.cab_matrix(Q, cin, cout)
    obp = orc.OracleBP(og, Q, 0)
    obp.init_messages(1, beliefs, tc, orc.Rng(3))
    obp.set_params(cab, na, 1.0)
    od = [obp.sweep_sync(1.0) for _ in range(4)]
    opsi, omsg = obp.get_state()
    for mode in (0, 1):
        bp = S.bp_conditional()
        bp.init_messages(S.blockmodel_t(g, Q, 0), 1, beliefs, tc, 3)
        bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
        bp.set_gather_mode(mode)
        d = [bp.sweep(1, 1.0) for _ in range(4)]
        psi, msg = bp.get_state()
        assert np.abs(np.array(d) - np.array(od)).max() < 1e-9, (mode, d, od)
        assert np.abs(psi - opsi).max() < 1e-9 and np.abs(msg - omsg).max() < 1e-9, mode
        assert psi[hub, 1] == 1.0 and psi[7, 0] == 1.0


def test_degree_corrected_powerlaw_graph_c4_family(S, orc):
    """config C4 family at test size: DC-SBM with power-law propensities, Q=8, --deg_corr_flag 1; rows
    above the segment capacity go through the hub kernels, everything is checked against the oracle"""
    from sbm_bp_amd import synth
    N, Q = 30000, 8
    pairs, cab, c_eff = synth.dc_sbm_powerlaw(N, Q, 8.0, 0.1, 3)
    g = S.Graph.from_edges(pairs, N)
    og = orc.Graph.from_edges(pairs, N)
    assert g.max_degree > 512 and 6.0 < c_eff < 10.0
    tc = synth.true_conf(N, Q)
    na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)
    bp = S.bp_conditional()
    bp.init_messages(S.blockmodel_t(g, Q, 1), 0, None, tc, 2)
    bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
    assert bp.stats().n_hub_rows >= 1
    obp = orc.OracleBP(og, Q, 1)
    obp.init_messages(0, None, tc, orc.Rng(2))
    obp.set_params(cab, na, 1.0)
    for _ in range(4):
        assert abs(bp.sweep(1, 1.0) - obp.sweep_sync(1.0)) < 1e-9
    psi, msg = bp.get_state()
    opsi, omsg = obp.get_state()
    # rows with ~1000 factors: products in a different order differ by ~1e-11 (north star: 1e-5)
    assert np.abs(msg - omsg).max() < 1e-9 and np.abs(psi - opsi).max() < 1e-9
    niter, last = bp.converge(1e-10, 2000, 1.0)
    it2, _ = obp.converge_sync(1e-10, 2000, 1.0)
    assert niter >= 0 and abs((niter + 4) - (it2 + 4)) <= 1
    f, parts = bp.compute_free_energy(parts=True)
    of, oparts = obp.free_energy(0)
    assert np.abs(parts - oparts).max() < 1e-9 * max(1.0, np.abs(oparts).max())
    assert abs(bp.compute_overlap() - obp.overlap()) < 1e-9
    assert bp.compute_overlap() > 0.5  # the planted groups are recovered


@pytest.mark.parametrize("pairs,N", [([], 7), ([[0, 1]], 2), ([[0, 1], [1, 2], [5, 6]], 9)])
def test_degenerate_graphs(S, orc, pairs, N):
    """no edges at all, a single edge, isolated vertices: psi_i = normalise(eta * F) for degree-0 rows"""
    Q = 3
    arr = np.array(pairs, dtype=np.uint32).reshape(-1, 2)
    g = S.Graph.from_edges(arr, N)
    og = orc.Graph.from_edges(arr, N)
    tc = (np.arange(N) % Q).astype(np.uint32)
    cab = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, 0.7], [0.5, 0.7, 5.0]]) * min(1.0, N / 10.0)  # keep cab < N (1 - cab/N > 0)
    na = np.array([max(1, N // Q)] * Q, dtype=np.uint32)
    bp = S.bp_conditional()
    bp.init_messages(S.blockmodel_t(g, Q, 0), 0, None, tc, 4)
    bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
    obp = orc.OracleBP(og, Q, 0)
    obp.init_messages(0, None, tc, orc.Rng(4))
    obp.set_params(cab, na, 1.0)
    niter, last = bp.converge(1e-13, 500, 1.0)
    it2, _ = obp.converge_sync(1e-13, 500, 1.0)
    # the marginal-gather form starts its exact 1-step check once the 2-step hint fires: when BP converges within
    # a sweep or two, niter can exceed the explicit form's by one; never less, never a false stop
    assert it2 <= niter <= it2 + 1 and last < 1e-13
    psi, msg = bp.get_state()
    opsi, omsg = obp.get_state()
    assert np.abs(psi - opsi).max() < 1e-13 and (msg.size == 0 or np.abs(msg - omsg).max() < 1e-13)
    f, parts = bp.compute_free_energy(parts=True)
    of, oparts = obp.free_energy(0)
    assert np.abs(parts - oparts).max() < 1e-12
    assert abs(bp.compute_entropy() - obp.entropy(0)[0]) < 1e-11
    assert abs(bp.compute_overlap() - obp.overlap()) < 1e-13


def test_q8_learning_follows_the_synchronous_oracle(S, orc):
    from sbm_bp_amd import synth
    N, Q = 4000, 8
    pairs, cin, cout = synth.planted_partition(N, Q, 14.0, 0.05, 21)
    g = S.Graph.from_edges(pairs, N)
    og = orc.Graph.from_edges(pairs, N)
    tc = synth.true_conf(N, Q)
    cab0 = synth.cab_matrix(Q, 0.8 * cin, 1.5 * cout)
    na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)
    bm = S.blockmodel_t(g, Q, 0)
    bp = S.bp_basic()
    bp.init_messages(bm, 0, None, tc, 9)
    res = bp.learning(bm, S.bp_blockmodel_state(cab0, na), 1e-6, 60, 0.2, 1.0)
    obp = orc.OracleBP(og, Q, 0)
    obp.init_messages(0, None, tc, orc.Rng(9))
    obp.set_params(cab0, na, 1.0)
    steps, f = obp.learning(1e-6, 60, 0.2, 1.0, None, sync=True, series_K=0)
    cab, na1 = bp.get_params()
    ocab, ona = obp.get_params()
    assert res.em_steps == steps and list(na1) == list(ona)
    assert np.abs(cab - ocab).max() < 1e-6 * np.abs(ocab).max() and abs(res.free_energy - f) < 1e-8
    assert res.overlap > 0.9


def test_clamped_rows_run_in_the_marginal_gather_form(S, orc):
    """-i 1 (bp_conditional with one-hot clamped rows): after the first explicit sweep the engine uses the marginal-gather
    kernel and still follows the oracle's synchronous sweeps; an arbitrary state (set_state) falls back to message gather"""
    a = args_of(golden("c1_planted_i1_seed0"))
    _, _, bp, _ = engine_from(S, a)
    _, obp, _ = oracle_from(orc, a)
    for _ in range(5):
        assert abs(bp.sweep(1, 1.0) - obp.sweep_sync(1.0)) < 1e-12
    psi, msg = bp.get_state()
    opsi, omsg = obp.get_state()
    assert np.abs(psi - opsi).max() < 1e-12 and np.abs(msg - omsg).max() < 1e-12
    assert bp.stats().psi_form_sweeps == 4
    clamped = np.flatnonzero(np.asarray(a["beliefs"]) != -1)
    assert (psi[clamped].max(1) == 1.0).all()  # untouched one-hot marginals
    bp.set_state(psi, msg)
    before = bp.stats().psi_form_sweeps
    bp.sweep(3, 1.0)
    assert bp.stats().psi_form_sweeps == before


def _full_size_oracle_parity(S, orc, pairs, N, Q, dc, cab, seed, row_tol=5e-12, hub_tol=None, sweeps=3):
    """the engine against the synchronous oracle at a BASELINE configuration's FULL size: `sweeps` sweeps from the same seeded
    state, messages and marginals compared entry for entry, then the free energy by the series (orders as the engine picks).
    row_tol: the worst of 1e6..1e7 rows after three sweeps of the marginal-gather form (one more division per edge than the
    oracle's message gather) sits at 2e-12; the max difference per sweep agrees to 1e-12."""
    from sbm_bp_amd import synth
    g = S.Graph.from_edges(pairs, N)
    og = orc.Graph.from_edges(pairs, N)
    del pairs
    tc = synth.true_conf(N, Q)
    na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)
    bp = S.bp_conditional()
    bp.init_messages_device(S.blockmodel_t(g, Q, dc), tc, seed)
    bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
    psi0, msg0 = bp.get_state()
    obp = orc.OracleBP(og, Q, dc)
    obp.init_messages(0, None, tc, orc.Rng(0))
    obp.set_state(psi0, msg0)
    obp.set_params(cab, na, 1.0)
    del psi0, msg0
    for k in range(sweeps):
        d1, d2 = bp.sweep(1, 1.0), obp.sweep_sync(1.0)
        assert abs(d1 - d2) < max(1e-12, 0.1 * row_tol), (k, d1, d2)
    assert bp.stats().psi_form_sweeps == sweeps  # the marginal-gather kernel (first sweep straight from the device state)
    psi, msg = bp.get_state()
    opsi, omsg = obp.get_state()
    deg = og.deg
    long_rows = np.flatnonzero(deg > 32)
    tol_rows = np.full(N, row_tol)
    if hub_tol is not None:
        tol_rows[long_rows] = hub_tol  # wave / workgroup products multiply in another order than the oracle's row loop
    assert (np.abs(psi - opsi).max(1) <= tol_rows).all(), float(np.abs(psi - opsi).max())
    src = np.repeat(np.arange(N), deg)
    assert (np.abs(msg - omsg).max(1) <= tol_rows[src]).all(), float(np.abs(msg - omsg).max())
    assert np.abs(psi.sum(1) - 1).max() < 1e-14
    del psi, msg, opsi, omsg, src
    obp.compute_h()
    f, parts = bp.compute_free_energy(parts=True)
    of, oparts = obp.free_energy(3)
    assert np.abs(parts - oparts).max() < 1e-10 * max(1.0, np.abs(oparts).max()), (parts, oparts)
    na_e, nna_e, cab_e = bp.em_expectations()
    ona, onna, ocab = obp.em_expect()
    assert np.abs(na_e - ona).max() < 1e-9 * N and np.abs(cab_e - ocab).max() < 1e-9 * max(1.0, np.abs(ocab).max())
    assert abs(bp.compute_overlap() - obp.overlap()) < 1e-12
    return g


def test_full_size_oracle_parity_c2(S, orc):
    """BASELINE C2 (N=1e6, Q=2, c=3) at full size against the oracle: index widths, the XCD-padded grid and the two-stage
    fold only show at scale"""
    from sbm_bp_amd import synth
    pairs, cin, cout = synth.planted_partition(1_000_000, 2, 3.0, 0.1, 1)
    _full_size_oracle_parity(S, orc, pairs, 1_000_000, 2, 0, synth.cab_matrix(2, cin, cout), 42)


def test_full_size_oracle_parity_c5(S, orc):
    """BASELINE C5 (N=1e6, Q=4, c=5) at full size against the oracle"""
    from sbm_bp_amd import synth
    pairs, cin, cout = synth.planted_partition(1_000_000, 4, 5.0, 0.1, 4)
    _full_size_oracle_parity(S, orc, pairs, 1_000_000, 4, 0, synth.cab_matrix(4, cin, cout), 7)


def test_full_size_oracle_parity_c4(S, orc):
    """BASELINE C4 (power-law DC-SBM, N=1e6, Q=8, --deg_corr_flag 1, rows up to ~2500 edges) at full size against the oracle:
    segment, wave-product and hub kernels in one sweep. 1e-9: rows above 32 edges multiply hundreds of factors in another order
    than the oracle's row loop (measured 7e-11), and from the second sweep on their neighbours inherit that difference"""
    from sbm_bp_amd import synth
    pairs, cab, _ = synth.dc_sbm_powerlaw(1_000_000, 8, 8.0, 0.1, 3)
    g = _full_size_oracle_parity(S, orc, pairs, 1_000_000, 8, 1, cab, 99, row_tol=1e-9, hub_tol=1e-9)
    assert g.max_degree > 1000


@pytest.mark.skipif(os.environ.get("SBMBP_FULL_C3_ORACLE", "1") == "0", reason="switched off (SBMBP_FULL_C3_ORACLE=0); ~2 min of oracle time, 20 GB of host memory")
def test_full_size_oracle_parity_c3(S, orc):
    """the headline configuration C3 (N=1e7, Q=4, c=10: 1e8 directed edges) at full size against the oracle, two sweeps
    (on by default since round 3: the GPU tier has the time)"""
    from sbm_bp_amd import synth
    pairs, cin, cout = synth.planted_partition(10_000_000, 4, 10.0, 0.1, 2)
    _full_size_oracle_parity(S, orc, pairs, 10_000_000, 4, 0, synth.cab_matrix(4, cin, cout), 1234, sweeps=2)
