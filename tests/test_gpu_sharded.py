"""GPU tier for the multi-GPU driver (csrc/dist.hip): the ranks of a run as threads of this process on one GPU
(LocalShards: in-process transport), checked against one rank, the single engine and the reference goldens; every
sweep form (marginal gather, message gather with damping / clamped rows / dc 2); hub rows; three processes over gloo
(callback transport) and RCCL with one rank."""
import numpy as np
import pytest

from conftest import args_of, best_perm_diff, best_vector_perm, golden

pytestmark = pytest.mark.gpu


def _problem(orc, name):
    gd = golden(name)
    a, r = args_of(gd), gd["result"]
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    bp = orc.OracleBP(g, a["Q"], a["dc"])
    bp.init_messages(0, None, a["true_conf"], orc.Rng(a["seed"]))
    cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    psi0, msg0 = bp.get_state()
    return a, r, g, cab, na, psi0, msg0


def _sharded(g, a, cab, na, psi0, msg0, world, n_chunks=None):
    from sbm_bp_amd.distributed import LocalShards
    sb = LocalShards.from_csr(g.row_ptr, g.nbr, a["Q"], a["dc"], world, n_chunks)
    sb.init_messages_device(7, a["true_conf"])  # labels (true_conf) ...
    sb.set_state_global(psi0, msg0)             # ... and the reference's initial state
    sb.expand_bp_params(cab, na, a["beta"])
    return sb


@pytest.mark.parametrize("name,world", [("c1_matched_tight_seed0", 2), ("c1_matched_tight_seed0", 5), ("q4_tight_seed0", 3),
                                        ("c1_dc1_tight_seed0", 4), ("q10_tight_seed1", 3)])
def test_sharded_equals_unsharded_and_reference(orc, name, world):
    a, r, g, cab, na, psi0, msg0 = _problem(orc, name)
    one = _sharded(g, a, cab, na, psi0, msg0, 1)
    sb = _sharded(g, a, cab, na, psi0, msg0, world)
    for _ in range(2):
        d1, dk = one.sweep(3), sb.sweep(3)
        assert abs(d1 - dk) < (1e-13 if a["Q"] <= 8 else 1e-11)
        psi_k, msg_k = sb.global_state()
        psi_1, msg_1 = one.global_state()
        # partition invariance: only the reduction order of the Q field sums differs; the far-from-converged
        # Q = 10 transient amplifies that last-bit difference a little more than the others
        tol = 1e-12 if a["Q"] <= 8 else 1e-11
        assert np.abs(psi_k - psi_1).max() < tol and np.abs(msg_k - msg_1).max() < tol
    niter, exact = sb.converge(1e-12, 3000, 1.0, check_every=6)
    assert niter >= 0 and exact < 1e-12
    assert one.converge(1e-12, 3000, 1.0, check_every=1)[0] == niter
    psi = sb.global_state()[0]
    # the first sweep after set_state gathers the messages themselves (cut edges shipped), as on the single engine: the
    # shards follow the single engine's trajectory and land in the reference's basin (also at Q = 10, several fixed points)
    d, _ = best_perm_diff(psi, np.array(r["psi"]).reshape(psi.shape))
    assert d < 1e-9  # the reference's fixed point
    if a["Q"] <= 8:
        assert abs(sb.compute_overlap() - r["overlap"]) < 1e-9
    assert np.abs(psi - one.global_state()[0]).max() < 1e-9
    assert sb.stats()[0].psi_form_sweeps > 0


@pytest.mark.parametrize("world", [2, 5])
def test_adaptive_relaxation_is_taken_identically_on_every_rank(S, orc, world):
    """hub_dc0 (plain Jacobi oscillates): every rank folds the same all-gathered sums, so all of them lower the field mix on the
    same sweep without talking about it, and the sharded run lands where the single engine and the reference do"""
    a, r, g, cab, na, psi0, msg0 = _problem(orc, "hub_dc0_tight_seed0")
    one = _sharded(g, a, cab, na, psi0, msg0, 1)
    sb = _sharded(g, a, cab, na, psi0, msg0, world)
    n1, l1 = one.converge(a["crit"], a["tmax"], 1.0)
    nk, lk = sb.converge(a["crit"], a["tmax"], 1.0, check_every=5)
    assert n1 >= 0 and nk == n1 and one.relaxation() == sb.relaxation() == (1, -1)
    psi = sb.global_state()[0]
    assert np.abs(psi - one.global_state()[0]).max() < 1e-10
    assert best_perm_diff(psi, np.array(r["psi"]).reshape(psi.shape))[0] < 1e-9
    assert abs(sb.compute_free_energy() - r["f"]) < 1e-9 * abs(r["f"])
    sb.close()
    one.close()


def test_sharded_matches_single_engine_fixed_point(S, orc):
    a, r, g, cab, na, psi0, msg0 = _problem(orc, "q4_tight_seed0")
    sb = _sharded(g, a, cab, na, psi0, msg0, 3)
    gg = S.load_edge_list(a["path"], a["N"])
    bp = S.bp_conditional()
    bp.init_messages(S.blockmodel_t(gg, a["Q"], 0), 0, None, a["true_conf"], a["seed"])
    bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
    # same state, same schedule: the shards reproduce the single engine sweep for sweep (first sweep message gather, then
    # marginal gather), not just its fixed point
    for _ in range(3):
        assert abs(sb.sweep(2) - bp.sweep(2, 1.0)) < 1e-13
        assert np.abs(sb.global_state()[0] - bp.real_psi()).max() < 1e-12
    n1, _ = sb.converge(1e-12, 3000, 1.0)
    n2, _ = bp.converge(1e-12, 3000, 1.0)
    assert n1 == n2 and np.abs(sb.global_state()[0] - bp.real_psi()).max() < 1e-11


@pytest.fixture(scope="module")
def S():
    import sbm_bp_amd as S
    S.load_library()
    return S


def _hub_graph(N, hub_deg, seed):
    rng = np.random.default_rng(seed)
    e = [np.stack([np.zeros(hub_deg, dtype=np.int64), rng.choice(np.arange(1, N), hub_deg, replace=False)], 1),
         np.stack([np.full(hub_deg - 300, 1, dtype=np.int64), rng.choice(np.arange(2, N), hub_deg - 300, replace=False)], 1),
         rng.integers(0, N, size=(3 * N, 2))]
    return np.concatenate(e).astype(np.uint32)


@pytest.mark.parametrize("Q,dc", [(3, 1), (2, 0), (4, 1)])
def test_hub_rows_both_forms_and_oracle(S, orc, Q, dc):
    """rows above the segment capacity (512 edges, 1024 for Q=2) take the workgroup-per-row kernels"""
    N = 6000
    pairs = _hub_graph(N, 1700, 3)
    g = S.Graph.from_edges(pairs, N)
    og = orc.Graph.from_edges(pairs, N)
    assert g.max_degree > 1024
    tc = (np.arange(N) * Q // N).astype(np.uint32)
    cab = np.full((Q, Q), 0.004 if dc else 1.0) + np.eye(Q) * (0.02 if dc else 5.0)
    na = np.array([N // Q] * Q, dtype=np.uint32)
    res = []
    for mode in (0, 1):
        bp = S.bp_conditional()
        bp.init_messages(S.blockmodel_t(g, Q, dc), 0, None, tc, 11)
        bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
        bp.set_gather_mode(mode)
        assert bp.stats().n_hub_rows >= 2
        d = [bp.sweep(1, 1.0) for _ in range(4)]
        res.append((d, bp.get_state(), bp.compute_free_energy(parts=True)[1], bp.stats().psi_form_sweeps))
    obp = orc.OracleBP(og, Q, dc)
    obp.init_messages(0, None, tc, orc.Rng(11))
    obp.set_params(cab, na, 1.0)
    od = [obp.sweep_sync(1.0) for _ in range(4)]
    opsi, omsg = obp.get_state()
    obp.compute_h()
    _, oparts = obp.free_energy(0)
    assert res[0][3] == 3 and res[1][3] == 0
    for d, (psi, msg), parts, _ in res:
        assert np.abs(np.array(d) - np.array(od)).max() < 1e-11
        assert np.abs(psi - opsi).max() < 1e-11 and np.abs(msg - omsg).max() < 1e-11
        assert np.abs(parts - oparts).max() < 1e-9 * max(1.0, np.abs(oparts).max())


@pytest.mark.parametrize("Q,dc,world", [(3, 1, 3), (2, 0, 2)])
def test_sharded_with_hub_rows(S, orc, Q, dc, world):
    """hub rows (one workgroup each) inside the chunks of a shard: sweeps, convergence and free energy equal the single shard"""
    from sbm_bp_amd.distributed import LocalShards
    N = 6000
    pairs = _hub_graph(N, 1700, 3)
    g = S.Graph.from_edges(pairs, N)
    tc = (np.arange(N) * Q // N).astype(np.uint32)
    cab = np.full((Q, Q), 0.004 if dc else 1.0) + np.eye(Q) * (0.02 if dc else 5.0)
    na = np.array([N // Q] * Q, dtype=np.uint32)
    runs = []
    for w in (1, world):
        sb = LocalShards(g, Q, dc, w, n_chunks=4)
        sb.init_messages_device(5, tc)
        sb.expand_bp_params(cab, na, 1.0)
        assert sum(st.n_hub_rows for st in sb.stats()) >= 2
        d = [sb.sweep(1) for _ in range(4)]
        psi, msg = sb.global_state()
        runs.append((d, psi, msg, sb.compute_free_energy(), sb.compute_overlap()))
        sb.close()
    (d1, p1, m1, f1, o1), (dk, pk, mk, fk, ok) = runs
    assert np.abs(np.array(d1) - np.array(dk)).max() < 1e-12
    assert np.abs(p1 - pk).max() < 1e-12 and np.abs(m1 - mk).max() < 1e-12
    assert abs(f1 - fk) < 1e-10 * max(1.0, abs(f1)) and abs(o1 - ok) < 1e-12


@pytest.mark.parametrize("name,world", [("c1_matched_tight_seed0", 3), ("q4_tight_seed0", 2), ("c1_dc1_tight_seed0", 4)])
def test_sharded_reductions_equal_single_engine_and_reference(S, orc, name, world):
    """free energy, entropy, EM expectations over shards (all-reduced partials) vs the single engine with the
    same non-edge evaluation (moment series), and vs the reference golden"""
    a, r, g, cab, na, psi0, msg0 = _problem(orc, name)
    sb = _sharded(g, a, cab, na, psi0, msg0, world)
    res = sb.inference(1e-13, 4000, 1.0)
    assert res["niter"] >= 0
    gg = S.load_edge_list(a["path"], a["N"])
    bp = S.bp_basic()
    bp.init_messages(S.blockmodel_t(gg, a["Q"], a["dc"]), 0, None, a["true_conf"], a["seed"])
    bp.expand_bp_params(S.bp_blockmodel_state(cab, na))
    bp.converge(1e-13, 4000, 1.0)
    f1, p1 = bp.compute_free_energy(parts=True)  # exact all-pairs non-edge term at this N, on the shards as well
    fk, pk = sb.compute_free_energy(parts=True)
    assert np.abs(pk - p1).max() < 1e-10 * max(1.0, np.abs(p1).max())
    assert abs(res["free_energy"] - r["f"]) < 2e-9 * max(1.0, abs(r["f"]))  # the reference golden (exact O(N^2) loop there too)
    bound = 0.0
    e1, q1 = bp.compute_entropy(parts=True)
    ek, qk = sb.compute_entropy(parts=True)
    if a["dc"]:
        assert np.isnan(ek) and np.isnan(e1)
    else:
        assert np.abs(qk - q1).max() < 1e-10 * max(1.0, np.abs(q1).max())
        assert abs(ek - r["e"]) < max(2e-8, 20 * bound) * max(1.0, abs(r["e"]))
    na1, nna1, cab1 = bp.em_expectations()
    nak, nnak, cabk = sb.em_expectations()
    p = best_vector_perm(nak, na1)
    assert np.abs(nak[p] - na1).max() < 1e-7 and np.abs(nnak[p] - nna1).max() < 1e-6
    assert np.abs(cabk[np.ix_(p, p)] - cab1).max() < 1e-8 * max(1.0, np.abs(cab1).max())
    assert abs(res["overlap"] - bp.compute_overlap()) < 1e-9


def test_sharded_learning_matches_single_engine_and_reference(S, orc):
    """-m learn over 3 shards: the EM run of the single engine step for step (same schedule: a message-gather sweep after
    every parameter change), hence the reference's learned parameters to the same 1e-7"""
    gd = golden("c1_learn_515_seed0")
    a, r = args_of(gd), gd["result"]
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    obp = orc.OracleBP(g, a["Q"], 0)
    obp.init_messages(0, None, a["true_conf"], orc.Rng(a["seed"]))
    cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    psi0, msg0 = obp.get_state()
    sb = _sharded(g, a, cab, na, psi0, msg0, 3)
    res = sb.learning(a["lcrit"], a["tmax"], a["lr"], 1.0)
    gg = S.load_edge_list(a["path"], a["N"])
    bm = S.blockmodel_t(gg, a["Q"], 0)
    bp = S.bp_basic()
    bp.init_messages(bm, 0, None, a["true_conf"], a["seed"])
    one = bp.learning(bm, S.bp_blockmodel_state(cab, na), a["lcrit"], a["tmax"], a["lr"], 1.0)
    cab1, na1 = bp.get_params()
    assert res["status"] == 1 and one.status == 1 and res["em_steps"] == one.em_steps
    assert list(res["na"]) == list(na1) and np.abs(res["cab"] - cab1).max() < 1e-8 * np.abs(cab1).max()
    assert abs(res["overlap"] - one.overlap) < 1e-9 and abs(res["free_energy"] - one.free_energy) < 1e-10
    ref_cab = np.array(r["cab_final"]).reshape(2, 2)
    assert list(res["na"]) == list(r["na_final"]) and np.abs(res["cab"] - ref_cab).max() < 1e-7 * np.abs(ref_cab).max()
    assert abs(res["overlap"] - r["overlap"]) < 1e-7


def test_message_gather_modes_on_shards(S, orc):
    """everything the marginal-gather form cannot do runs sharded through the message-gather form (cut-edge messages shipped
    every sweep): damping, clamped rows (-i 1), deg_corr_flag 2, a zero in cab. Same iterates as the single engine."""
    from sbm_bp_amd.distributed import LocalShards
    cases = [("c1_matched_damped_seed0", {}), ("c1_planted_i1_seed0", {}), ("c1_dc2_tight_seed0", {}), ("q4_tight_seed0", {"zero": True})]
    for name, opt in cases:
        gd = golden(name)
        a = args_of(gd)
        gg = S.load_edge_list(a["path"], a["N"])
        if "eps" in a:
            st = S.bp_param_from_epsilon_c(S.blockmodel_t(gg, a["Q"], a["dc"]), a["eps"], a["c"])
        else:
            st = S.bp_param_from_direct(S.blockmodel_t(gg, a["Q"], a["dc"]), a["pa"], a["cab_upper"])
        cab = st.cab.copy()
        if opt.get("zero"):
            cab[0, 1] = cab[1, 0] = 0.0
        conf = a.get("beliefs") if a["init_flag"] else None
        bp = S.bp_conditional()
        bp.init_messages(S.blockmodel_t(gg, a["Q"], a["dc"]), a["init_flag"], conf, a["true_conf"], a["seed"])
        bp.set_beta(a["beta"])
        bp.expand_bp_params(S.bp_blockmodel_state(cab, st.na))
        sb = LocalShards(gg, a["Q"], a["dc"], 3, n_chunks=2)
        sb.init_messages(a["init_flag"], conf, a["true_conf"], a["seed"], True)
        sb.expand_bp_params(cab, st.na, a["beta"])
        for _ in range(3):
            d1, dk = bp.sweep(2, a["damp"]), sb.sweep(2, a["damp"])
            assert abs(d1 - dk) < 1e-13, name
            p1, m1 = bp.get_state()
            pk, mk = sb.global_state()
            assert np.abs(pk - p1).max() < 1e-12 and np.abs(mk - m1).max() < 1e-12, name
        n1, l1 = bp.converge(1e-10, 3000, a["damp"])
        nk, lk = sb.converge(1e-10, 3000, a["damp"])
        assert n1 == nk and (n1 >= 0 or opt.get("zero")), name
        f1, fk = bp.compute_free_energy(parts=True)[1], sb.compute_free_energy(parts=True)[1]
        assert np.abs(f1 - fk).max() < 1e-10 * max(1.0, np.abs(f1).max()), name
        e1, ek = bp.em_expectations(), sb.em_expectations()
        assert np.abs(e1[2] - ek[2]).max() < 1e-9 * max(1.0, np.abs(e1[2]).max()), name
        assert abs(bp.compute_overlap() - sb.compute_overlap()) < 1e-12, name
        sb.close()


def test_reference_stream_initial_state_on_shards(S):
    """sbmbp_dist_init_messages: every rank draws the reference's std::mt19937 stream and keeps its slice"""
    from sbm_bp_amd.distributed import LocalShards
    a = args_of(golden("q4_tight_seed0"))
    gg = S.load_edge_list(a["path"], a["N"])
    psi, msg = gg.initial_state(a["Q"], 0, None, a["seed"])
    sb = LocalShards(gg, a["Q"], 0, 4)
    sb.init_messages(0, None, a["true_conf"], a["seed"], True)
    pk, mk = sb.global_state()
    assert (pk == psi).all() and np.abs(mk - msg).max() < 1e-15  # the one restored component of a record: a few ulp
    sb.close()


def test_rccl_transport_with_one_rank(S, orc):
    """the RCCL communicators themselves (ncclCommInitRank from an id, all-gather, all-reduce) run with the one rank a
    one-GPU box allows; the multi-rank exchange is the same grouped ncclSend/ncclRecv code with more peers"""
    from sbm_bp_amd.capi import COMM_ID_BYTES, check
    from sbm_bp_amd.distributed import Comm, ShardedBP
    import ctypes as C
    lib = S.load_library()
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    check(lib.sbmbp_comm_unique_id(buf))
    h = C.c_void_p()
    check(lib.sbmbp_comm_init_rank(C.byref(h), bytes(buf), 1, 0, 0))
    comm = Comm(h.value)
    assert comm.transport == "rccl" and comm.world == 1
    a, r, g, cab, na, psi0, msg0 = _problem(orc, "q4_tight_seed0")
    sb = ShardedBP(S.Graph.from_csr(g.row_ptr, g.nbr), a["Q"], 0, comm)
    sb.init_messages_device(7, a["true_conf"])
    sb.set_state(psi0, msg0)
    sb.expand_bp_params(cab, na, 1.0)
    res = sb.inference(1e-12, 3000, 1.0)
    assert res["niter"] >= 0 and abs(res["free_energy"] - r["f"]) < 2e-9 * max(1.0, abs(r["f"]))
    sb.close()
    # several drivers one after the other on the SAME communicators, with different chunk counts: what bench.py --gpus N does
    # when it measures the chunk count (one plan per candidate, the losers closed)
    first = None
    for nc in (1, 4, 2):
        sb = ShardedBP(S.Graph.from_csr(g.row_ptr, g.nbr), a["Q"], 0, comm, 0, nc)
        assert sb.info.n_chunks == nc
        sb.init_messages_device(7, a["true_conf"])
        sb.set_state(psi0, msg0)
        sb.expand_bp_params(cab, na, 1.0)
        d = sb.sweep(4, 1.0)
        psi, _ = sb.get_state()
        if first is None:
            first = (d, psi)
        else:
            assert abs(d - first[0]) < 1e-14 and np.abs(psi - first[1]).max() < 1e-14
        sb.close()


@pytest.mark.parametrize("name,world", [("q4_tight_seed0", 3)])
def test_three_processes_share_one_gpu(orc, tmp_path, name, world):
    """the multi-PROCESS path: one rank per process, every process on cuda:0, the C++ driver with the callback transport
    (gloo between the processes; RCCL refuses two ranks on one device, tools/probe_nccl_dup.py). Same calls, same order,
    same chunking as the RCCL path; result = the run with the ranks as threads."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = tmp_path / "result.npz"
    from bench import free_port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "sharded_gpu_worker.py"), str(out), name]
    pr = subprocess.run(cmd, env=env, timeout=600, capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-3000:]
    res = np.load(out)
    a, r, g, cab, na, psi0, msg0 = _problem(orc, name)
    ref = _sharded(g, a, cab, na, psi0, msg0, world)
    assert abs(ref.sweep(3) - float(res["d3"])) < 1e-15
    niter, exact = ref.converge(1e-12, 3000, 1.0, check_every=6)
    assert int(res["niter"]) == niter
    psi = ref.global_state()[0]
    assert np.abs(res["psi"] - psi).max() < 1e-14
    assert abs(float(res["overlap"]) - ref.compute_overlap()) < 1e-14
    assert abs(float(res["fe"]) - ref.compute_free_energy()) < 1e-13
    assert abs(float(res["entropy"]) - ref.compute_entropy()) < 1e-13
    # the reference's fixed point (non-edge term exact at this N, on shards too)
    assert abs(float(res["fe"]) - r["f"]) < 2e-9 * max(1.0, abs(r["f"]))


def test_two_processes_two_gpus_over_rccl(S, orc, tmp_path):
    """the production transport with more than one rank: two processes, one GPU each, the library's own RCCL communicators
    (grouped ncclSend/ncclRecv per chunk, all-gather of the fold rows). Needs two devices: skipped on the one-GPU boxes this
    repository is developed on, there for the first node that has them. Result = the run with the ranks as threads."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    if S.load_library().sbmbp_device_count() < 2:
        pytest.skip("needs two GPUs")
    from bench import free_port
    name, world = "q4_tight_seed0", 2
    out = tmp_path / "result.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "sharded_gpu_worker.py"), str(out), name, "rccl"]
    pr = subprocess.run(cmd, env=env, timeout=600, capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-3000:]
    res = np.load(out)
    a, r, g, cab, na, psi0, msg0 = _problem(orc, name)
    ref = _sharded(g, a, cab, na, psi0, msg0, world)
    assert abs(ref.sweep(3) - float(res["d3"])) < 1e-15
    niter, exact = ref.converge(1e-12, 3000, 1.0, check_every=6)
    assert int(res["niter"]) == niter
    assert np.abs(res["psi"] - ref.global_state()[0]).max() < 1e-14
    assert abs(float(res["fe"]) - ref.compute_free_energy()) < 1e-13 and abs(float(res["fe"]) - r["f"]) < 2e-9 * max(1.0, abs(r["f"]))


def test_bench_multi_rank_line_from_a_bare_shell():
    """`python bench.py --gpus 2` started as a plain process on this one-GPU box (SBMBP_REHEARSAL=1: both ranks on cuda:0, the C++
    driver over the callback transport): it launches its own ranks, measures the chunk count (1/2/4/8) in its set-up, and rank 0
    prints ONE line whose fields the driver reads; the single-engine line of the same workload says the same about the graph"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, SBMBP_REHEARSAL="1")
    env.pop("SBMBP_SHARD_CHUNKS", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    pr = subprocess.run(cmd, env=env, timeout=600, capture_output=True, text=True, cwd=ROOT)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["metric"] == "BP edge-message updates/sec" and d["higher_is_better"] is True and d["dtype"] == "f64" and d["vs_baseline"] is None
    trials = d["config"]["chunk_trials_ms_per_sweep"]
    assert sorted(trials) == ["1", "2", "4", "4s", "8"] and all(v > 0 for v in trials.values())  # (round 2's "4p" is an opt-in now)
    assert d["config"]["chunk_choice"] in trials and str(d["config"]["exchange"]["chunks"]) == d["config"]["chunk_choice"].rstrip("ps")
    assert d["config"]["exchange"]["transport"] == "callbacks"
    assert d["converge"]["converged"] is True and d["converge"]["overlap"] > 0.9
    assert abs(d["value"] - d["steps"] * d["config"]["E2"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                            env=env, timeout=600, capture_output=True, text=True, cwd=ROOT)
    assert single.returncode == 0, single.stderr[-3000:]
    s1 = json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][-1])
    assert s1["n_gpus"] == 1 and s1["config"]["E2"] == d["config"]["E2"]
    assert s1["converge"]["sweeps"] == d["converge"]["sweeps"] and abs(s1["converge"]["overlap"] - d["converge"]["overlap"]) < 1e-9


def test_full_size_c3_two_shards_equal_one(S):
    """the headline configuration (N=1e7, Q=4, c=10) cut into two shards with the chunked exchange and the fused send/receive
    buffers: same iterates as one shard, seen through reductions (max difference, overlap, free energy, row sums)"""
    from sbm_bp_amd import synth
    from sbm_bp_amd.distributed import LocalShards
    N, Q = 10_000_000, 4
    pairs, cin, cout = synth.planted_partition(N, Q, 10.0, 0.1, 2)
    g = S.Graph.from_edges(pairs, N)
    del pairs
    tc = synth.true_conf(N, Q)
    cab, na = synth.cab_matrix(Q, cin, cout), np.array([N // Q] * Q, dtype=np.uint32)
    out = []
    for w in (1, 2):
        sb = LocalShards(g, Q, 0, w, n_chunks=4 if w > 1 else 1)
        sb.init_messages_device(1234, tc)
        sb.expand_bp_params(cab, na, 1.0)
        d = [sb.sweep(1) for _ in range(3)]
        out.append((d, sb.compute_overlap(), sb.compute_free_energy(), sb.confusion()))
        if w > 1:
            assert sb.ranks[0].n_halo > 3_000_000  # nearly every vertex is a boundary vertex at c = 10
        sb.close()
    (d1, o1, f1, r1), (d2, o2, f2, r2) = out
    assert np.abs(np.array(d1) - np.array(d2)).max() < 1e-12 and abs(o1 - o2) < 1e-12
    assert abs(f1 - f2) < 1e-10 * abs(f1) and np.abs(r1 - r2).max() < 1e-9 * N
